#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], "C2"): procedural Sponza-class atrium(seed=1), 262,144 triangles, 1920x1080,
myMaxRecursionDepth 4 (<= 5 rays per sample), Sky.cpp atmosphere on. One STEP = one pass of the whole hot path
(generate -> [extend -> shade] x 5 -> sky -> resolve) over one batch of --spp-per-step samples per pixel; scene, BVH and
all queues are resident in HBM before the timed region. With N GPUs every rank renders the full frame for its own
sample indices (weak scaling) and the timed region ends with one RCCL reduce of the accumulation targets.

Prints ONE JSON line on rank 0: whole-job Mrays/s (+ Msamples/s), the extend kernel's roofline object and the CPU
oracle baseline (bounded sample, rank 0 at N=1 only).
"""
import argparse
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
PMC_ROUND = "r02"                                  # profiles/<round>/pmc_traffic.json: counters of THIS build
DEF_DEFER, DEF_POSTPONE = 4, 40                   # the library defaults (csrc/kernels.h LaunchCfg)
# k_extend's own VALU roof: most of its vector instructions (min/max, compares, selects, shifts, byte permutes, v_fma_mix)
# issue at HALF rate on gfx950 (tools/valurate -> profiles/r02/valu_issue_rates.txt); weighted over its node and triangle
# steps that is 3.52 cycles per wave64 instruction per SIMD (tools/valu_mix.py -> profiles/r02/valu_mix_extend.txt)
VALU_MIX_CYCLES_EXTEND = 3.52
VALU_MIX_ROOF_GINSTR = 256 * 4 * 2.4 / VALU_MIX_CYCLES_EXTEND


def make_workload(name):
    from pathtracer_amd import scenes
    if name == "atrium":
        return scenes.atrium(1), scenes.ATRIUM_CAM, 4, "C2 atrium(seed=1) 262144 tris, sky on"
    if name == "street":
        return scenes.street(2), scenes.STREET_CAM, 8, "C4 street(seed=2) 3145728 tris, sky on"
    if name == "courtyard":
        return scenes.courtyard(3), scenes.COURTYARD_CAM, 16, "C5 courtyard(seed=3) 10485760 tris, sky on"
    if name == "cornell":
        return scenes.cornell_box(), None, 4, "C1 Cornell box 36 tris, sky off"
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=64,
                    help="samples per pixel rendered by one step (one pt_render call = one wavefront batch)")
    ap.add_argument("--depth", type=int, default=-1, help="myMaxRecursionDepth (default: the workload's)")
    ap.add_argument("--workload", default="atrium", choices=["atrium", "street", "courtyard", "cornell"])
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
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

    import numpy as np
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
    scene, cam, depth, label = make_workload(args.workload)
    if args.depth >= 0:
        depth = args.depth
    if args.material_ext:
        from pathtracer_amd import scenes as _scn
        scene = _scn.with_material_ext(scene)
        label += ", GGX + procedural albedo (extension)"

    pt = PathTracer(local_rank, W, H)
    pt.myHalfResRender = False
    pt.UpdateOutputTexture()
    from pathtracer_amd import scenes as _sc
    if cam is None:
        pt.LoadScene(scene, _sc.CORNELL_CAM_POS)
        pt.mySampleSky = False
        pt.mySkyFallbackIntensity = 0.0
    else:
        pt.LoadScene(scene, cam["eye"])
        pt.myCamera.LookAt(cam["eye"], cam["target"])
        pt.mySampleSky = True
        pt.myLightInstanceIdx = -1
        pt.mySky.mySunDir = np.array([0.3, 0.8, 0.52], dtype=np.float32) / np.float32(np.linalg.norm([0.3, 0.8, 0.52]))
    pt.myMaxRecursionDepth = depth
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

    total_steps = WU + K
    # this rank's sample indices: a contiguous block of (WU+K)*SPP samples
    first0 = rank * total_steps * SPP

    def step(i):
        ctx.render(consts, W, H, first0 + i * SPP, SPP)

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for i in range(WU):
        step(i)
    if world > 1:   # warm the collective path too (communicator/channel set-up is lazy), on a scratch tensor
        scratch = torch.zeros_like(target)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        del scratch
    barrier()
    ctx.stats_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(WU + i)
    ctx.synchronize()
    if world > 1:   # the one exchange step of the job: sum of the per-GPU float targets onto rank 0
        _, n_total = ptdist.reduce_sum_to_root(target, (WU + K) * SPP)
    barrier()
    t1 = time.perf_counter()
    st = ctx.stats()
    used_packet = st.packet_launches > 0   # bounce 0 went through k_extend_packet (PT_OPT_PRIMARY_PACKET, library's choice at 2)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    counts = torch.tensor([st.rays, st.samples], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed_s = float(elapsed.item())
    rays_all, samples_all = float(counts[0].item()), float(counts[1].item())

    roofline = None
    if rank == 0 and not args.no_roofline:
        # (1) algorithmic bytes: counting instantiation of the extend kernel on one step of the same workload, run with
        #     the CANONICAL schedule (nearest-first, every triangle group tested at once: no postponing / deferral), so
        #     the figure does not grow when a faster schedule visits a few more nodes. The shipped schedule's own
        #     counts are reported beside it.
        def count_pass(defer, postpone, primaries_only=False):
            ctx.set_option(api.PT_OPT_EXTEND_DEFER, defer)
            ctx.set_option(api.PT_OPT_EXTEND_POSTPONE_PCT, postpone)
            ctx.stats_reset()
            ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 1)   # (the counting pass traces bounce 0 with k_extend too)
            if primaries_only:
                pt.myMaxRecursionDepth = 0
                ctx.render(pt.BuildRtConsts(), W, H, first0 + total_steps * SPP, SPP)
                pt.myMaxRecursionDepth = depth
            else:
                step(total_steps)
            r = ctx.stats()
            ctx.set_option(api.PT_OPT_COUNT_TRAVERSAL, 0)
            return r
        sc_ = count_pass(0, 0)
        ship = count_pass(DEF_DEFER, DEF_POSTPONE)
        cn, ct, cr = sc_.nodes_visited, sc_.tris_tested, sc_.rays
        sn, st_, sr, snw, stw = ship.nodes_visited, ship.tris_tested, ship.rays, ship.node_wave_steps, ship.tri_wave_steps
        if used_packet and depth > 0:
            # bounce 0 belongs to k_extend_packet: k_extend's per-ray figures are those of the secondary bounces (same samples
            # traced to depth 0, subtracted)
            p0, p1 = count_pass(0, 0, True), count_pass(DEF_DEFER, DEF_POSTPONE, True)
            cn, ct, cr = cn - p0.nodes_visited, ct - p0.tris_tested, cr - p0.rays
            sn, st_, sr, snw, stw = sn - p1.nodes_visited, st_ - p1.tris_tested, sr - p1.rays, snw - p1.node_wave_steps, stw - p1.tri_wave_steps
        nodes_per_ray = cn / max(1, cr)
        tris_per_ray = ct / max(1, cr)
        bytes_per_ray = NODE_BYTES * nodes_per_ray + TRI_BYTES * tris_per_ray + RAY_IO_BYTES
        # (2) extend-kernel time: HIP events on the ctx stream around every extend launch, K more steps
        ctx.stats_reset()
        ctx.set_option(api.PT_OPT_TIME_KERNELS, 1)
        for i in range(K):
            step(total_steps + 1 + i)
        se = ctx.stats()
        ctx.set_option(api.PT_OPT_TIME_KERNELS, 0)
        ext_s = se.extend_ms * 1e-3
        # (3) counters that cannot be read from inside this process (rocprofv3 --pmc passes of this same command,
        #     tools/prof_pmc.sh -> tools/pmc_traffic.py -> profiles/r02/pmc_traffic.json, matched on workload, size,
        #     batch and depth): VALU wave-instructions and fabric-side bytes per extend launch of THIS build.
        pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", PMC_ROUND, "pmc_traffic.json")) as f:
                for e in json.load(f):
                    if (e["workload"], e["width"], e["height"], e["spp_per_step"], e["depth"]) == (args.workload, W, H, SPP, depth) \
                            and (e.get("material_ext", 0), e.get("russian_roulette", 0)) == (args.material_ext, args.russian_roulette):
                        pmc = e
        except (OSError, ValueError, KeyError):
            pass
        # k_extend's launches: every bounce, or bounces >= 1 when the fused primaries went through k_extend_packet
        packet = se.packet_launches > 0
        b_from = 1 if packet else 0
        k_launches = max(1, sum(se.extend_launches_bounce[b_from:]))
        k_ms = sum(se.extend_ms_bounce[b_from:])
        k_rays = sum(se.rays_bounce[b_from:])
        launches = k_launches
        launch_s = k_ms / launches * 1e-3
        alg_bytes = bytes_per_ray * k_rays / launches
        alg_flop = (NODE_FLOP * nodes_per_ray + TRI_FLOP * tris_per_ray) * k_rays / launches
        hbm = {"algorithmic_bytes_per_launch": round(alg_bytes, 1), "achieved_gbs": round(alg_bytes / launch_s / 1e9, 2),
               "peak_gbs": HBM_PEAK_GBS, "ratio_to_hbm_peak": round(alg_bytes / launch_s / 1e9 / HBM_PEAK_GBS, 4),
               "note": "SURVEY 8(d) B_ray x rays; the BVH is served from L1/L2/Infinity Cache when it fits (C2 14.7 MB, C4 176 MB), "
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
            valu = {"wave_instr_per_launch": pmc["extend_valu_wave_instr_per_launch"],
                    "lanes_active_per_instr": pmc.get("extend_lanes_active_per_valu_instr"),
                    "achieved_ginstr_per_s": round(g, 1), "peak_ginstr_per_s": VALU_PEAK_GINSTR, "frac": round(g / VALU_PEAK_GINSTR, 4),
                    "mix_cycles_per_instr": VALU_MIX_CYCLES_EXTEND, "mix_roof_ginstr_per_s": round(VALU_MIX_ROOF_GINSTR, 1),
                    "frac_of_mix_roof": round(min(1.0, g / VALU_MIX_ROOF_GINSTR), 4),
                    "note": "peak = every instruction at the full rate (2 cycles per wave64 instruction per SIMD); the kernel's own mix is mostly "
                            "half-rate instructions (profiles/%s/valu_issue_rates.txt, valu_mix_extend.txt), hence the mix roof" % PMC_ROUND}
        # vector-memory pipe (L1 -> VGPR return path): every wave-level load moves 64 lanes x its width whatever the lanes
        # do with it; steps of the shipped schedule (COUNT pass) x loads per step (node: 5 x 16 B, triangle: 16+12+12 B)
        l1_bytes = (snw * 64.0 * 80.0 + stw * 64.0 * 40.0) / max(1, sr) * k_rays / launches
        vmem = {"l1_return_bytes_per_launch": round(l1_bytes, 1), "achieved_gbs": round(l1_bytes / launch_s / 1e9, 1),
                "peak_gbs": L1_PEAK_GBS, "frac": round(l1_bytes / launch_s / 1e9 / L1_PEAK_GBS, 4),
                "ta_busy_frac": pmc.get("extend_ta_busy_frac") if pmc else None, "td_busy_frac": pmc.get("extend_td_busy_frac") if pmc else None,
                "ta_stalled_by_cache_frac": pmc.get("extend_ta_stalled_by_tc_frac") if pmc else None,
                "note": "64 B/clk/CU x 256 CUs x 2.4 GHz; busy fractions from the PMC profile (cycles with work queued, stalls on cache misses included)"}
        # The binding roof is the larger measured fraction: VALU issue (PMC instruction count / live launch time) or
        # fabric-side traffic against the HBM peak. Without a PMC profile of this build + workload the line falls back to
        # the algorithmic fp32 rate against the vector peak (never above 1).
        if valu and (not fabric or valu["frac"] >= fabric["frac_of_hbm_peak"]):
            top = {"bound": "valu", "achieved": valu["achieved_ginstr_per_s"], "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s", "frac": valu["frac"],
                   "frac_of_mix_roof": valu["frac_of_mix_roof"], "mix_roof": valu["mix_roof_ginstr_per_s"],
                   "basis": "SQ_INSTS_VALU per launch (profiles/%s/pmc_traffic.json) / live HIP-event launch time; peak = 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles" % PMC_ROUND}
        elif fabric:
            top = {"bound": "hbm", "achieved": fabric["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fabric["frac_of_hbm_peak"],
                   "basis": "measured fabric-side bytes per launch (profiles/%s/pmc_traffic.json) / live HIP-event launch time" % PMC_ROUND}
        else:
            top = {"bound": "valu", "achieved": flops["achieved_tflops"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops["frac"],
                   "basis": "algorithmic fp32 flops (no PMC profile committed for this workload/size/build)"}
        roofline = {
            **top, "kernel": "k_extend", "traffic": traffic,
            "hbm_algorithmic": hbm, "fabric": fabric, "valu_issue": valu, "vmem_pipe": vmem, "flops_algorithmic": flops,
            "nodes_per_ray": round(nodes_per_ray, 3), "tris_per_ray": round(tris_per_ray, 3),
            "shipped_schedule": {"nodes_per_ray": round(sn / max(1, sr), 3),
                                 "tris_per_ray": round(st_ / max(1, sr), 3),
                                 "node_phase_lane_util": round(sn / max(1, 64 * snw), 3),
                                 "tri_phase_lane_util": round(st_ / max(1, 64 * stw), 3)},
            "bytes_per_ray": round(bytes_per_ray, 1),
            "launches": int(k_launches), "avg_launch_ms": round(k_ms / launches, 4),
            "primary_kernel": ({"kernel": "k_extend_packet", "launches": int(se.extend_launches_bounce[0]),
                                "avg_launch_ms": round(se.extend_ms_bounce[0] / max(1, se.extend_launches_bounce[0]), 4),
                                "mrays_per_s": round(se.rays_bounce[0] / max(se.extend_ms_bounce[0], 1e-9) / 1e3, 1),
                                "note": "bounce 0 (coherent 8x8-pixel tiles): one BVH walk per wave, scalar node / triangle fetches"}
                               if packet else None),
            "primary_kernel_choice": {"option": args.primary_packet, "timed_ms_k_extend": round(se.packet_cal_ms[0], 4),
                                      "timed_ms_k_extend_packet": round(se.packet_cal_ms[1], 4)},
            "bytes_per_launch": round(alg_bytes, 1),
            "extend_share_of_render": round(se.extend_ms / max(se.render_ms, 1e-9), 4),
            "extend_mrays_per_s": round(se.rays / max(ext_s, 1e-12) / 1e6, 2),
            "extend_mrays_per_s_by_bounce": [round(se.rays_bounce[b] / max(se.extend_ms_bounce[b], 1e-9) / 1e3, 1)
                                             for b in range(16) if se.rays_bounce[b] > 0],
            "secondary_mrays_per_s": round(sum(se.rays_bounce[1:]) / max(sum(se.extend_ms_bounce[1:]), 1e-9) / 1e3, 1),
        }

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

    if rank == 0:
        out = {
            "metric": "Mrays/s + Msamples/s at 1080p, 4 bounces, 1/2/4/8 MI355X",
            "value": round(rays_all / elapsed_s / 1e6, 2),
            "unit": "Mrays/s",
            "msamples_per_s": round(samples_all / elapsed_s / 1e6, 2),
            "n_gpus": world, "steps": K, "warmup": WU,
            "ms_per_step": round(elapsed_s / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "max_recursion_depth": depth, "spp_per_step": SPP,
                       "sky_mode": args.sky_mode, "pipeline": args.pipeline,
                       "material_ext": args.material_ext, "russian_roulette": args.russian_roulette,
                       "triangles": int(info.num_triangles), "bvh_nodes": int(info.num_bvh_nodes),
                       "bvh_mb": round(info.bvh_bytes / 1e6, 2), "bvh_build_ms": round(info.build_ms, 1),
                       "rays_per_sample": round(rays_all / max(samples_all, 1), 3),
                       "parallelism": f"sample-partition x{world}" if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
