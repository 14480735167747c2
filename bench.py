#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Headline workload (BASELINE.json configs[1], "C2"): procedural Sponza-class atrium(seed=1), 262,144 triangles, 1920x1080,
myMaxRecursionDepth 4 (<= 5 rays per sample), Sky.cpp atmosphere on. One STEP = one pass of the whole hot path
(primaries -> [extend -> shade] x 5 -> sky -> resolve) over one batch of --spp-per-step samples per pixel; scene, BVH and
all queues are resident in HBM before the timed region.

N GPUs (one process per GPU): the K x spp-per-step samples of the job (or --total-spp, BASELINE C3: 4096) are
partitioned by sample index over the ranks (STRONG scaling, the default for N > 1; --weak gives every rank K steps of its
own), every rank renders the full frame for its samples, and the timed region ends with ONE RCCL reduce of the
accumulation targets.

Prints ONE JSON line on rank 0: whole-job Mrays/s (+ Msamples/s), the extend kernel's roofline object, the CPU oracle
baseline (bounded sample, rank 0 at N=1 only) and, at N=1, short legs of the other BASELINE configurations under
"configs" (C4 street, C4 with the GGX extension, the SAH-hostile street_hard stand-in, C5 courtyard at 4K).
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINSTR = 1228.8      # wave64 VALU instructions/s: 256 CUs x 4 SIMD32 x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md)
NODE_BYTES, TRI_BYTES, RAY_IO_BYTES = 80, 48, 48   # B_ray = 80*N_node + 48*N_tri + 48
NODE_FLOP, TRI_FLOP = 176.0, 45.0                  # F_ray = 8*22*N_node + 45*N_tri
FP32_PEAK_TFLOPS = 157.3                           # MI355X_MICROARCH.md: peak fp32 vector
L1_PEAK_GBS = 39321.6                              # vector L1 return path: 64 B/clk/CU x 256 CUs x 2.4 GHz
LEG_WARMUP = 5                                     # untimed steps of a "configs" leg (the library times its triangle phases on a scene's first five batches)
PMC_ROUNDS = ("r03", "r02")                        # profiles/<round>/pmc_traffic.json, newest first
DEF_DEFER, DEF_POSTPONE = 4, 40                    # the library defaults (csrc/kernels.h LaunchCfg)
# k_extend's own VALU roof: most of its vector instructions (min/max, selects, shifts, byte permutes, v_fma_mix) issue at
# HALF rate on gfx950 (tools/valurate -> profiles/r03/valu_issue_rates.txt); weighted over its node and triangle steps that
# is 3.41 cycles per wave64 instruction per SIMD (tools/valu_mix.py -> profiles/r03/valu_mix_extend.txt; 3.52 in round 2,
# before the hit-mask assembly of the box test moved to full-rate forms)
VALU_MIX_CYCLES_EXTEND = 3.41
VALU_MIX_ROOF_GINSTR = 256 * 4 * 2.4 / VALU_MIX_CYCLES_EXTEND

WORKLOADS = ("atrium", "street", "street_hard", "courtyard", "cornell")


def build_fingerprint():
    """sha256 over the sources the library is built from (csrc/*.{hip,inc,h,cpp}, its Makefile, include/*.h). A PMC entry of
    profiles/*/pmc_traffic.json carries the fingerprint of the build it was measured on (tools/pmc_traffic.py); counters of
    another build are not paired with this build's launch times."""
    h = hashlib.sha256()
    files = []
    for pat in ("pathtracer_amd/csrc/*.hip", "pathtracer_amd/csrc/*.inc", "pathtracer_amd/csrc/*.h", "pathtracer_amd/csrc/*.cpp",
                "pathtracer_amd/csrc/Makefile", "include/*.h"):
        files += glob.glob(os.path.join(ROOT, pat))
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def make_workload(name):
    from pathtracer_amd import scenes
    if name == "atrium":
        return scenes.atrium(1), scenes.ATRIUM_CAM, 4, "C2 atrium(seed=1) 262144 tris, sky on"
    if name == "street":
        return scenes.street(2), scenes.STREET_CAM, 8, "C4 street(seed=2) 3145728 tris, sky on"
    if name == "street_hard":
        return (scenes.street_hard(4), scenes.STREET_HARD_CAM, 8,
                "C4-hard street_hard(seed=4) 3145728 tris (yawed blocks, 7.4 % long thin triangles, rotated instanced clutter), sky on")
    if name == "courtyard":
        return scenes.courtyard(3), scenes.COURTYARD_CAM, 16, "C5 courtyard(seed=3) 10485760 tris, sky on"
    if name == "cornell":
        return scenes.cornell_box(), None, 4, "C1 Cornell box 36 tris, sky off"
    if name.startswith("obj:"):
        # "if $PT_SCENE_DIR/sponza.obj exists on the box, also run it" — any OBJ/MTL file through the Python host's
        # reader (scenes.load_obj). The camera is placed from the bounding box: a sixth of the height above the floor at the
        # centre, looking down the longer horizontal axis (the view the atrium stand-in uses for its own courtyard).
        import numpy as np
        path = name[4:]
        scene = scenes.load_obj(path)
        pts = np.concatenate([scene.meshes[m][0]["pos"] for m, _, _ in scene.instances])
        lo, hi = pts.min(axis=0).astype(np.float64), pts.max(axis=0).astype(np.float64)
        c, ext = 0.5 * (lo + hi), hi - lo
        eye = np.array([c[0], lo[1] + ext[1] / 6.0, c[2]])
        axis = 0 if ext[0] >= ext[2] else 2
        eye[axis] = c[axis] - 0.35 * ext[axis]
        target = eye.copy()
        target[axis] = c[axis] + 0.5 * ext[axis]
        cam = {"eye": tuple(float(x) for x in eye), "target": tuple(float(x) for x in target)}
        return scene, cam, 4, f"OBJ file {os.path.basename(path)} {scene.num_triangles} tris, sky on"
    raise ValueError(name)


def load_workload(pt, name, depth=-1, material_ext=0):
    """Scene + camera + sky of a workload into the host mirror; returns (scene, depth, label)."""
    import numpy as np
    from pathtracer_amd import scenes as _sc
    scene, cam, d, label = make_workload(name)
    if depth >= 0:
        d = depth
    if material_ext:
        scene = _sc.with_material_ext(scene)
        label += ", GGX + procedural albedo (extension)"
    if cam is None:
        pt.LoadScene(scene, _sc.CORNELL_CAM_POS)
        pt.mySampleSky = False
        pt.mySkyFallbackIntensity = 0.0
        pt.myLightInstanceIdx = 4
    else:
        pt.LoadScene(scene, cam["eye"])
        pt.myCamera.LookAt(cam["eye"], cam["target"])
        pt.mySampleSky = True
        pt.myLightInstanceIdx = -1
        pt.mySky.mySunDir = np.array([0.3, 0.8, 0.52], dtype=np.float32) / np.float32(np.linalg.norm([0.3, 0.8, 0.52]))
    pt.myMaxRecursionDepth = d
    return scene, d, label


def find_pmc(workload, W, H, SPP, depth, material_ext, russian_roulette, fingerprint, shared_phase=0):
    """The committed PMC entry of this exact workload and triangle-phase variant; (entry, stale). stale = measured on another
    build of the kernels."""
    for rnd in PMC_ROUNDS:
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as f:
                entries = json.load(f)
        except (OSError, ValueError):
            continue
        for e in entries:
            try:
                if (e["workload"], e["width"], e["height"], e["spp_per_step"], e["depth"]) == (workload, W, H, SPP, depth) \
                        and (e.get("material_ext", 0), e.get("russian_roulette", 0), e.get("shared_triangle_phase", 0)) == \
                        (material_ext, russian_roulette, shared_phase):
                    return dict(e, round=rnd), e.get("build_fingerprint") != fingerprint
            except KeyError:
                continue
    return None, False


def measure_extend(pt, ctx, api, step, W, H, SPP, depth, sample0, steps, leg):
    """Counting passes (canonical + shipped schedule) and a timed pass of `steps` steps with HIP events around every extend
    launch. Returns the roofline object of k_extend. leg: dict(workload, material_ext, russian_roulette, primary_packet)."""
    def count_pass(defer, postpone, primaries_only=False):
        ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, defer)
        ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, postpone)
        ctx.stats_reset()
        ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)   # (the counting pass traces bounce 0 with k_extend too)
        try:
            if primaries_only:
                pt.myMaxRecursionDepth = 0
                ctx.render(pt.BuildRtConsts(), W, H, sample0, SPP)
                pt.myMaxRecursionDepth = depth
            else:
                step(sample0)
            return ctx.stats()
        finally:
            ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
            ctx.set_tuning(api.PT_TUNE_EXTEND_DEFER, DEF_DEFER)
            ctx.set_tuning(api.PT_TUNE_EXTEND_POSTPONE_PCT, DEF_POSTPONE)
    # (1) extend-kernel time: HIP events on the ctx stream around every extend launch
    ctx.stats_reset()
    ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
    for i in range(steps):
        step(sample0 + (1 + i) * SPP)
    se = ctx.stats()
    ctx.set_option(api.PT_OPT_TIME_KERNELS, 0)
    packet = se.packet_launches > 0
    # (2) algorithmic bytes: counting instantiation of the extend kernel on one step of the same workload, run with the
    #     CANONICAL schedule (nearest-first, every triangle group tested at once: no postponing / deferral), so the figure
    #     does not grow when a faster schedule visits a few more nodes. The shipped schedule's own counts are reported
    #     beside it.
    sc_ = count_pass(0, 0)
    ship = count_pass(DEF_DEFER, DEF_POSTPONE)
    cn, ct, cr = sc_.nodes_visited, sc_.tris_tested, sc_.rays
    sn, st_, sr, snw, stw = ship.nodes_visited, ship.tris_tested, ship.rays, ship.node_wave_steps, ship.tri_wave_steps
    if packet and depth > 0:
        # bounce 0 belongs to k_extend_packet: k_extend's per-ray figures are those of the secondary bounces (same samples
        # traced to depth 0, subtracted)
        p0, p1 = count_pass(0, 0, True), count_pass(DEF_DEFER, DEF_POSTPONE, True)
        cn, ct, cr = cn - p0.nodes_visited, ct - p0.tris_tested, cr - p0.rays
        sn, st_, sr, snw, stw = sn - p1.nodes_visited, st_ - p1.tris_tested, sr - p1.rays, snw - p1.node_wave_steps, stw - p1.tri_wave_steps
    nodes_per_ray = cn / max(1, cr)
    tris_per_ray = ct / max(1, cr)
    bytes_per_ray = NODE_BYTES * nodes_per_ray + TRI_BYTES * tris_per_ray + RAY_IO_BYTES
    ext_s = se.extend_ms * 1e-3
    # (3) counters that cannot be read from inside this process (rocprofv3 --pmc passes of this same command,
    #     tools/prof_pmc.sh -> tools/pmc_traffic.py -> profiles/<round>/pmc_traffic.json, matched on workload, size, batch,
    #     depth AND the fingerprint of the kernel sources): VALU wave-instructions and fabric-side bytes per extend launch.
    fp = build_fingerprint()
    shared_used = 1 if se.shared_launches > 0 else 0
    pmc, stale = find_pmc(leg["workload"], W, H, SPP, depth, leg.get("material_ext", 0), leg.get("russian_roulette", 0), fp, shared_used)
    pmc_round = pmc["round"] if pmc else None
    if stale:
        pmc = None   # counters of another build are never paired with this build's launch times
    # k_extend's launches: every bounce, or bounces >= 1 when the fused primaries went through k_extend_packet
    b_from = 1 if packet else 0
    launches = max(1, sum(se.extend_launches_bounce[b_from:]))
    k_ms = sum(se.extend_ms_bounce[b_from:])
    k_rays = sum(se.rays_bounce[b_from:])
    launch_s = max(k_ms / launches * 1e-3, 1e-12)
    alg_bytes = bytes_per_ray * k_rays / launches
    alg_flop = (NODE_FLOP * nodes_per_ray + TRI_FLOP * tris_per_ray) * k_rays / launches
    hbm = {"algorithmic_bytes_per_launch": round(alg_bytes, 1), "achieved_gbs": round(alg_bytes / launch_s / 1e9, 2),
           "peak_gbs": HBM_PEAK_GBS, "ratio_to_hbm_peak": round(alg_bytes / launch_s / 1e9 / HBM_PEAK_GBS, 4),
           "note": "SURVEY 8(d) B_ray x rays; the BVH is served from L1/L2/Infinity Cache when it fits, "
                   "so this ratio can exceed 1 and HBM is then not the binding roof"}
    flops = {"algorithmic_flop_per_launch": round(alg_flop, 1), "achieved_tflops": round(alg_flop / launch_s / 1e12, 3),
             "peak_tflops": FP32_PEAK_TFLOPS, "frac": round(alg_flop / launch_s / 1e12 / FP32_PEAK_TFLOPS, 4),
             "note": "SURVEY 8(d): 176 flop per node visited + 45 per triangle tested, canonical schedule"}
    traffic = pmc["extend_bytes_per_launch"] if pmc else None
    fabric = ({"bytes_per_launch": traffic, "achieved_gbs": round(traffic / launch_s / 1e9, 2),
               "frac_of_hbm_peak": round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 4), "note": pmc["note"]} if traffic else None)
    valu = None
    if pmc and pmc.get("extend_valu_wave_instr_per_launch"):
        # the profiled command runs the same batch, so its per-launch instruction count belongs to these launches
        g = pmc["extend_valu_wave_instr_per_launch"] / launch_s / 1e9
        mix = g / VALU_MIX_ROOF_GINSTR
        valu = {"wave_instr_per_launch": pmc["extend_valu_wave_instr_per_launch"],
                "lanes_active_per_instr": pmc.get("extend_lanes_active_per_valu_instr"),
                "achieved_ginstr_per_s": round(g, 1), "peak_ginstr_per_s": VALU_PEAK_GINSTR, "frac": round(g / VALU_PEAK_GINSTR, 4),
                "mix_cycles_per_instr": VALU_MIX_CYCLES_EXTEND, "mix_roof_ginstr_per_s": round(VALU_MIX_ROOF_GINSTR, 1),
                "frac_of_mix_roof": round(mix, 4), "mix_roof_exceeded": bool(mix > 1.0),
                "note": "peak = every instruction at the full rate (2 cycles per wave64 instruction per SIMD); the kernel's own mix is mostly "
                        "half-rate instructions (profiles/r03/valu_issue_rates.txt, valu_mix_extend.txt), hence the mix roof; a value above 1 "
                        "means profile and build do not belong together and is flagged, not clamped"}
    # vector-memory pipe (L1 -> VGPR return path): every wave-level load moves 64 lanes x its width whatever the lanes
    # do with it; steps of the shipped schedule (COUNT pass) x loads per step (node: 5 x 16 B, triangle: 16+12+12 B)
    l1_bytes = (snw * 64.0 * 80.0 + stw * 64.0 * 40.0) / max(1, sr) * k_rays / launches
    vmem = {"l1_return_bytes_per_launch": round(l1_bytes, 1), "achieved_gbs": round(l1_bytes / launch_s / 1e9, 1),
            "peak_gbs": L1_PEAK_GBS, "frac": round(l1_bytes / launch_s / 1e9 / L1_PEAK_GBS, 4),
            "ta_busy_frac": pmc.get("extend_ta_busy_frac") if pmc else None, "td_busy_frac": pmc.get("extend_td_busy_frac") if pmc else None,
            "ta_stalled_by_cache_frac": pmc.get("extend_ta_stalled_by_tc_frac") if pmc else None,
            "note": "64 B/clk/CU x 256 CUs x 2.4 GHz; busy fractions from the PMC profile (cycles with work queued, stalls on cache misses included)"}
    # Top level. bound/achieved/peak/frac = the binding MEASURED roof (the larger of VALU issue and fabric traffic, from the
    # PMC profile of this very build); frac_algorithmic = SURVEY 8(d)'s figure for the same bound (algorithmic flops / fp32
    # vector peak, or algorithmic bytes / HBM peak), always present. Without a PMC profile of this build (none committed
    # for the workload, or pmc_stale) the top level IS the algorithmic figure.
    if valu and (not fabric or valu["frac"] >= fabric["frac_of_hbm_peak"]):
        top = {"bound": "valu", "achieved": valu["achieved_ginstr_per_s"], "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s", "frac": valu["frac"],
               "frac_algorithmic": flops["frac"], "frac_of_mix_roof": valu["frac_of_mix_roof"], "mix_roof": valu["mix_roof_ginstr_per_s"],
               "basis": "SQ_INSTS_VALU per launch (profiles/%s/pmc_traffic.json, same build fingerprint) / live HIP-event launch time; "
                        "peak = 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles; frac_algorithmic = SURVEY 8(d) flops / 157.3 TFLOP/s" % pmc_round}
    elif fabric:
        top = {"bound": "hbm", "achieved": fabric["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fabric["frac_of_hbm_peak"],
               "frac_algorithmic": hbm["ratio_to_hbm_peak"],
               "basis": "measured fabric-side bytes per launch (profiles/%s/pmc_traffic.json, same build fingerprint) / live HIP-event launch time; "
                        "frac_algorithmic = SURVEY 8(d) bytes / 8 TB/s (above 1 when the caches serve the BVH)" % pmc_round}
    else:
        top = {"bound": "valu", "achieved": flops["achieved_tflops"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops["frac"],
               "frac_algorithmic": flops["frac"],
               "basis": "SURVEY 8(d) algorithmic fp32 flops / vector peak (%s)" %
                        ("the committed PMC profile belongs to another build of the kernels: pmc_stale" if stale else "no PMC profile committed for this workload")}
    return {
        **top, "kernel": "k_extend", "traffic": traffic, "pmc_stale": bool(stale), "build_fingerprint": fp,
        "hbm_algorithmic": hbm, "fabric": fabric, "valu_issue": valu, "vmem_pipe": vmem, "flops_algorithmic": flops,
        "nodes_per_ray": round(nodes_per_ray, 3), "tris_per_ray": round(tris_per_ray, 3),
        "shipped_schedule": {"nodes_per_ray": round(sn / max(1, sr), 3),
                             "tris_per_ray": round(st_ / max(1, sr), 3),
                             "node_phase_lane_util": round(sn / max(1, 64 * snw), 3),
                             "tri_phase_lane_util": round(st_ / max(1, 64 * stw), 3)},
        "bytes_per_ray": round(bytes_per_ray, 1),
        "launches": int(launches), "avg_launch_ms": round(k_ms / launches, 4),
        "primary_kernel": ({"kernel": "k_extend_packet", "launches": int(se.extend_launches_bounce[0]),
                            "avg_launch_ms": round(se.extend_ms_bounce[0] / max(1, se.extend_launches_bounce[0]), 4),
                            "mrays_per_s": round(se.rays_bounce[0] / max(se.extend_ms_bounce[0], 1e-9) / 1e3, 1),
                            "note": "bounce 0 (one pixel's samples per wave): one BVH walk per wave, scalar node / triangle fetches"}
                           if packet else None),
        "primary_kernel_choice": {"option": leg.get("primary_packet", 2), "timed_ms_k_extend": round(se.packet_cal_ms[0], 4),
                                  "timed_ms_k_extend_packet": round(se.packet_cal_ms[1], 4)},
        "triangle_phase": {"option": leg.get("shared_triangle_phase", 2), "choice": int(se.shared_choice), "shared_used": shared_used,
                           "timed_ms_per_lane": round(se.shared_cal_ms[0], 4), "timed_ms_shared": round(se.shared_cal_ms[1], 4),
                           "note": "PT_OPT_SHARED_TRIANGLE_PHASE: the wave pools its rays' pending leaf records and tests them 64 at a time; "
                                   "option 2 times two batches of each form, alternating, during the warm-up steps (the secondary bounces of a scene's second to fifth batch; the times below are the sums) and keeps the shared one if it was faster by 1.5 %"},
        "bytes_per_launch": round(alg_bytes, 1),
        "extend_share_of_render": round(se.extend_ms / max(se.render_ms, 1e-9), 4),
        "extend_mrays_per_s": round(se.rays / max(ext_s, 1e-12) / 1e6, 2),
        "extend_mrays_per_s_by_bounce": [round(se.rays_bounce[b] / max(se.extend_ms_bounce[b], 1e-9) / 1e3, 1)
                                         for b in range(16) if se.rays_bounce[b] > 0],
        "extend_ms_share_by_bounce": [round(se.extend_ms_bounce[b] / max(se.extend_ms, 1e-9), 4)
                                      for b in range(16) if se.rays_bounce[b] > 0],
        "rays_share_by_bounce": [round(se.rays_bounce[b] / max(se.rays, 1), 5) for b in range(16) if se.rays_bounce[b] > 0],
        "secondary_mrays_per_s": round(sum(se.rays_bounce[1:]) / max(sum(se.extend_ms_bounce[1:]), 1e-9) / 1e3, 1),
    }


def run_config_leg(pt, ctx, api, ptdist, dev, spec):
    """One short leg of another BASELINE configuration on the same context: load, warm up, time `steps` steps, roofline."""
    t_leg = time.perf_counter()
    W, H, SPP, steps = spec["width"], spec["height"], spec["spp_per_step"], spec["steps"]
    pt.myWindowWidth, pt.myWindowHeight = W, H                # LoadScene takes the camera's aspect from the window
    scene, depth, label = load_workload(pt, spec["workload"], -1, spec.get("material_ext", 0))
    info = ctx.scene_info()
    target = ptdist.bind_torch_target(ctx, W, H, dev)
    pt.mySky.ComputeTranmittanceLut()
    consts = pt.BuildRtConsts()
    ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
    ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, spec.get("russian_roulette", 0))
    ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, spec.get("shared_triangle_phase", 2))

    def step(first):
        ctx.render(consts, W, H, first, SPP)
    for i in range(LEG_WARMUP):  # warm-up: the one-off primary-kernel timing of this scene / frame size, and the five batches the
        step(i * SPP)            # library needs to time its triangle phases (PT_OPT_SHARED_TRIANGLE_PHASE = 2)
    ctx.synchronize()
    ctx.stats_reset()
    t0 = time.perf_counter()
    for i in range(steps):
        step((LEG_WARMUP + i) * SPP)
    ctx.synchronize()
    t1 = time.perf_counter()
    st = ctx.stats()
    roof = measure_extend(pt, ctx, api, step, W, H, SPP, depth, (LEG_WARMUP + 1 + steps) * SPP, steps, spec)
    ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, 0)
    ctx.synchronize()
    del target
    el = t1 - t0
    return {
        "workload": label, "width": W, "height": H, "max_recursion_depth": depth, "spp_per_step": SPP,
        "material_ext": spec.get("material_ext", 0), "russian_roulette": spec.get("russian_roulette", 0),
        "steps": steps, "ms_per_step": round(el / steps * 1e3, 3),
        "mrays_per_s": round(st.rays / el / 1e6, 2), "msamples_per_s": round(st.samples / el / 1e6, 2),
        "secondary_mrays_per_s": roof["secondary_mrays_per_s"], "extend_mrays_per_s": roof["extend_mrays_per_s"],
        "nodes_per_ray": roof["nodes_per_ray"], "tris_per_ray": roof["tris_per_ray"], "bytes_per_ray": roof["bytes_per_ray"],
        "rays_per_sample": round(st.rays / max(1, st.samples), 3),
        "triangles": int(info.num_triangles), "bvh_nodes": int(info.num_bvh_nodes), "bvh_mb": round(info.bvh_bytes / 1e6, 2),
        "bvh_build_ms": round(info.build_ms, 1),
        "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "pmc_stale", "traffic", "avg_launch_ms",
                                          "launches", "shipped_schedule", "primary_kernel", "triangle_phase")},
        "hbm_algorithmic_ratio": roof["hbm_algorithmic"]["ratio_to_hbm_peak"],
        "leg_wall_s": round(time.perf_counter() - t_leg, 1),
    }


CONFIG_LEGS = (
    ("C4", {"workload": "street", "width": 1920, "height": 1080, "spp_per_step": 64, "steps": 3}),
    ("C4_ggx", {"workload": "street", "width": 1920, "height": 1080, "spp_per_step": 64, "steps": 2, "material_ext": 1}),
    ("C4_hard", {"workload": "street_hard", "width": 1920, "height": 1080, "spp_per_step": 64, "steps": 2}),
    ("C5", {"workload": "courtyard", "width": 3840, "height": 2160, "spp_per_step": 16, "steps": 2, "russian_roulette": 1}),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=64,
                    help="samples per pixel rendered by one step (one pt_render call = one wavefront batch)")
    ap.add_argument("--depth", type=int, default=-1, help="myMaxRecursionDepth (default: the workload's)")
    ap.add_argument("--workload", default="atrium", choices=list(WORKLOADS) + ["obj"])
    ap.add_argument("--scene-obj", default="", help="--workload obj: the OBJ/MTL file to render (default: $PT_SCENE_DIR/sponza.obj)")
    ap.add_argument("--sky-mode", type=int, default=0, choices=[0, 1],
                    help="0 = ray-march the sky on every miss (reference semantics); 1 = direction LUT (relMSE ~1e-6)")
    ap.add_argument("--material-ext", type=int, default=0, choices=[0, 1],
                    help="1 = attach the GGX + procedural-albedo material extension (BASELINE C4/C5 'extension' variant; the "
                         "reference's own materials are Lambert + Phong, the default)")
    ap.add_argument("--russian-roulette", type=int, default=0, choices=[0, 1], help="PT_OPT_RUSSIAN_ROULETTE (C5 variant)")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1],
                    help="1 = two pipeline lanes per batch (PT_OPT_PIPELINE): higher throughput, but extend launches then "
                         "overlap other kernels and the per-launch durations of the roofline grow")
    ap.add_argument("--primary-packet", type=int, default=2, choices=[0, 1, 2],
                    help="PT_OPT_PRIMARY_PACKET: 0 = the primaries are traced by k_extend, 1 = by k_extend_packet, 2 (library default) = "
                         "by whichever the library times faster on this scene and frame size")
    ap.add_argument("--shared-triangle-phase", type=int, default=2, choices=[0, 1, 2],
                    help="PT_OPT_SHARED_TRIANGLE_PHASE: 0 = every lane tests its own ray's leaf records, 1 = the wave pools them, 2 (library "
                         "default) = whichever the library times faster on this scene (during the warm-up step)")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="N > 1, strong scaling: samples per pixel of the whole job, partitioned over the ranks (default: steps x "
                         "spp-per-step; BASELINE C3 is 4096)")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling instead (every rank renders --steps steps of its own samples)")
    ap.add_argument("--bit-identical", action="store_true",
                    help="strong scaling through the 8 fixed virtual partitions (distributed.render_virtual_partitions): the image is the "
                         "same bits for 1, 2, 4 and 8 ranks; prints its sha256 as image_sha256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other BASELINE configurations (\"configs\" in the JSON line)")
    ap.add_argument("--configs", default="", help="comma-separated subset of the legs to run: " + ",".join(k for k, _ in CONFIG_LEGS) +
                    " (and C2_sponza_obj when $PT_SCENE_DIR/sponza.obj or --scene-obj names a file)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run), BEFORE
    # anything touches torch or the GPU in this process, and hand back the children's exit code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from pathtracer_amd import PathTracer, api
    from pathtracer_amd import distributed as ptdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.all_ranks_on_gpu0:
        local_rank = 0
    if world != max(1, args.gpus):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
              f"(python bench.py --gpus N does it by itself)", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    W, H, K, WU, SPP = args.width, args.height, args.steps, args.warmup, args.spp_per_step
    obj_path = args.scene_obj or (os.path.join(os.environ["PT_SCENE_DIR"], "sponza.obj") if os.environ.get("PT_SCENE_DIR") else "")
    if args.workload == "obj":
        if not os.path.isfile(obj_path):
            print(f"bench.py: --workload obj needs --scene-obj FILE or $PT_SCENE_DIR/sponza.obj ({obj_path!r} is not a file)", file=sys.stderr)
            sys.exit(2)
        args.workload = "obj:" + obj_path
    pt = PathTracer(local_rank, W, H)
    pt.myHalfResRender = False
    pt.UpdateOutputTexture()
    scene, depth, label = load_workload(pt, args.workload, args.depth, args.material_ext)
    ctx = pt.ctx
    info = ctx.scene_info()
    target = ptdist.bind_torch_target(ctx, W, H, dev)   # accumulation target RCCL can reduce in place
    if pt.mySampleSky:
        pt.mySky.ComputeTranmittanceLut()
        pt._sky_dirty = False
    consts = pt.BuildRtConsts()
    ctx.set_option(api.PT_OPT_SAMPLES_PER_BATCH, SPP)
    ctx.set_option(api.PT_OPT_SKY_MODE, args.sky_mode)
    ctx.set_option(api.PT_OPT_PIPELINE, args.pipeline)
    ctx.set_option(api.PT_OPT_RUSSIAN_ROULETTE, args.russian_roulette)
    ctx.set_option(api.PT_OPT_PRIMARY_PACKET, args.primary_packet)
    if args.shared_triangle_phase != 2 or not os.environ.get("PT_AMD_LIB"):   # (an experiment library may predate the option)
        ctx.set_option(api.PT_OPT_SHARED_TRIANGLE_PHASE, args.shared_triangle_phase)

    # ---- which samples this rank renders in the timed region ----
    strong = world > 1 and not args.weak
    total_spp = args.total_spp if args.total_spp > 0 else K * SPP
    if args.weak and world > 1:
        # every rank: K steps of SPP samples from its own contiguous block
        my_first, my_count = rank * K * SPP, K * SPP
        job_spp = world * K * SPP
    else:   # the job's samples (K x SPP, or --total-spp) partitioned over the ranks; at N = 1 the one rank takes them all
        my_first, my_count = ptdist.partition_samples(total_spp, world, rank)
        job_spp = total_spp
    warm_first = job_spp + rank * max(1, WU) * SPP          # warm-up samples: outside the job's sample range

    def render_range(first, count):
        done = 0
        while done < count:
            k = min(SPP, count - done)
            ctx.render(consts, W, H, first + done, k)
            done += k

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for i in range(WU):
        ctx.render(consts, W, H, warm_first + i * SPP, SPP)
    if world > 1:   # warm the collective path too (communicator/channel set-up is lazy), on a scratch tensor
        scratch = torch.zeros_like(target)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        if args.bit_identical:
            parts = [torch.empty_like(scratch) for _ in range(world)] if rank == 0 else None
            dist.gather(scratch, gather_list=parts, dst=0)
            del parts
        del scratch
    barrier()
    if WU:
        target = ptdist.bind_torch_target(ctx, W, H, dev)   # the warm-up samples do not belong to the job's image
    ctx.stats_reset()
    barrier()
    image_sha = None
    total = None
    t0 = time.perf_counter()
    if args.bit_identical:
        # 8 fixed virtual partitions whatever the rank count, block sums added on the root in partition order
        blocks = ptdist.render_virtual_partitions(ctx, consts, W, H, job_spp, world, rank, dev)
        total = ptdist.combine_virtual_partitions(blocks)
        n_total = job_spp
    else:
        render_range(my_first, my_count)
        ctx.synchronize()
        n_total = my_count
        if world > 1:   # the one exchange step of the job: sum of the per-GPU float targets onto rank 0
            _, n_total = ptdist.reduce_sum_to_root(target, my_count)
    barrier()
    t1 = time.perf_counter()
    if args.bit_identical and rank == 0:
        image_sha = hashlib.sha256(total.cpu().numpy().tobytes()).hexdigest()
    st = ctx.stats()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    counts = torch.tensor([st.rays, st.samples], dtype=torch.float64, device=dev)
    per_rank_ms = None
    if world > 1:
        allms = torch.zeros(world, dtype=torch.float64, device=dev)   # one slot per rank, summed: every backend has all_reduce
        allms[rank] = (t1 - t0) * 1e3
        dist.all_reduce(allms, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 3) for x in allms.tolist()]
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed_s = float(elapsed.item())
    rays_all, samples_all = float(counts[0].item()), float(counts[1].item())

    leg = {"workload": args.workload, "material_ext": args.material_ext, "russian_roulette": args.russian_roulette,
           "primary_packet": args.primary_packet, "shared_triangle_phase": args.shared_triangle_phase}
    roofline = None
    if rank == 0 and not args.no_roofline:
        if args.bit_identical:
            target = ptdist.bind_torch_target(ctx, W, H, dev)
        roofline = measure_extend(pt, ctx, api, lambda first: ctx.render(consts, W, H, first, SPP), W, H, SPP, depth,
                                  job_spp + (world + 1) * max(1, WU) * SPP, max(1, min(K, 8)), leg)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests import oracle_lib
        orc = oracle_lib.load()
        orc.set_russian_roulette(args.russian_roulette)   # the CPU leg runs the same variant of the workload
        oscene = orc.scene(scene)
        lut = orc.transmittance_lut(pt.mySky.myAtmosphereParams) if pt.mySampleSky else None
        # bounded sample of the same workload (same scene, camera, depth, sky): a short calibration pass at 1/4
        # resolution per axis sizes a full-resolution pass of whole samples to roughly 12 s of CPU work.
        c2 = pt.BuildRtConsts()
        _, cal, _ = oscene.render(c2, max(16, W // 4), max(16, H // 4), 0, 1, lut=lut)
        rate = cal["samples"] / max(cal["ns"] * 1e-9, 1e-6)
        cw, ch = W, H
        nspp = int(max(1, min(64, round(12.0 * rate / (W * H)))))
        if rate * 12.0 < W * H:   # very slow host: shrink the frame instead of dropping below 1 spp
            cw, ch, nspp = max(16, W // 2), max(16, H // 2), 1
        _, ostats, _ = oscene.render(c2, cw, ch, 0, nspp, lut=lut)
        sec = ostats["ns"] * 1e-9
        cpu_baseline = {
            "value": round(ostats["rays"] / sec / 1e6, 3), "unit": "Mrays/s", "cores": int(ostats["threads"]), "kind": "port",
            "msamples_per_s": round(ostats["samples"] / sec / 1e6, 3),
            "sample": f"{nspp} spp of the same workload at {cw}x{ch} ({ostats['samples']} samples, {ostats['rays']} rays, {sec:.2f} s), "
                      "oracle/libpt_oracle.so, OpenMP over rows",
        }
        orc.set_russian_roulette(0)
        del oscene

    # ---- short legs of the other BASELINE configurations (N = 1 only; the headline above is untouched by them) ----
    configs = None
    default_cmd = (args.workload == "atrium" and (W, H) == (1920, 1080) and not args.material_ext and not args.russian_roulette
                   and args.depth < 0 and not args.bit_identical)
    if rank == 0 and world == 1 and not args.no_configs and not args.no_roofline and (default_cmd or args.configs):
        legs = list(CONFIG_LEGS)
        if os.path.isfile(obj_path):   # a real asset on the box ($PT_SCENE_DIR/sponza.obj or --scene-obj) runs beside the stand-ins
            legs.append(("C2_sponza_obj", {"workload": "obj:" + obj_path, "width": 1920, "height": 1080, "spp_per_step": 64, "steps": 2}))
        want = [k for k in args.configs.split(",") if k] or [k for k, _ in legs]
        configs = {}
        ctx.set_option(api.PT_OPT_SKY_MODE, 0)
        ctx.set_option(api.PT_OPT_PIPELINE, 0)
        ctx.synchronize()
        del target
        for key, spec in legs:
            if key not in want:
                continue
            try:
                configs[key] = run_config_leg(pt, ctx, api, ptdist, dev, dict(spec, primary_packet=args.primary_packet,
                                                                              shared_triangle_phase=args.shared_triangle_phase))
            except Exception as e:   # a failed leg must not take the headline down with it; it is reported as failed
                configs[key] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        out = {
            "metric": "Mrays/s + Msamples/s at 1080p, 4 bounces, 1/2/4/8 MI355X",
            "value": round(rays_all / elapsed_s / 1e6, 2),
            "unit": "Mrays/s",
            "msamples_per_s": round(samples_all / elapsed_s / 1e6, 2),
            "n_gpus": world, "steps": K, "warmup": WU,
            "ms_per_step": round(elapsed_s / K * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "max_recursion_depth": depth, "spp_per_step": SPP,
                       "job_spp": int(job_spp), "spp_this_rank": int(my_count), "sky_mode": args.sky_mode, "pipeline": args.pipeline,
                       "material_ext": args.material_ext, "russian_roulette": args.russian_roulette,
                       "triangles": int(info.num_triangles), "bvh_nodes": int(info.num_bvh_nodes),
                       "bvh_mb": round(info.bvh_bytes / 1e6, 2), "bvh_build_ms": round(info.build_ms, 1),
                       "rays_per_sample": round(rays_all / max(samples_all, 1), 3),
                       "parallelism": (f"sample-partition x{world}, {'strong' if strong else 'weak'} scaling, one RCCL reduce"
                                       if world > 1 else "single GPU"),
                       "bit_identical": bool(args.bit_identical)},
            "per_rank_ms": per_rank_ms,
            "image_sha256": image_sha,
            "samples_reduced": int(n_total),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "configs": configs,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
