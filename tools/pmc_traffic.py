"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh into one entry of profiles/<PT_PROFILE_ROUND, default r03>/pmc_traffic.json.
Every entry carries build_fingerprint = bench.build_fingerprint() of the tree the passes ran on (sha256 of the kernel sources):
bench.py pairs an entry's counters with its live launch times only when the fingerprints agree, and prints pmc_stale otherwise.
usage: python tools/pmc_traffic.py <tag> <workload> <width> <height> <spp_per_step> <depth>
Units and corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are KB of L2<->fabric requests
(Infinity-Cache hits included); on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes for 16-byte-per-lane reads,
so it is doubled; the node/triangle gathers of k_extend are 16 bytes per lane but not streaming, i.e. outside the
calibrated pattern - the figure is an estimate and is labelled so."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402  (build_fingerprint)
from collections import defaultdict

tag, workload, w, h, spp, depth = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
acc, calls = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"gpurun_out/pmc_{tag}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
out = {"workload": workload, "width": w, "height": h, "spp_per_step": spp, "depth": depth, "build_fingerprint": bench.build_fingerprint(), "kernels": {}}
for k in acc:
    if "FETCH_SIZE" not in acc[k]:
        continue
    n = calls[k]["FETCH_SIZE"]
    fetch_kb = acc[k]["FETCH_SIZE"] / n
    write_kb = acc[k].get("WRITE_SIZE", 0.0) / max(1, calls[k].get("WRITE_SIZE", 1))
    valu = acc[k].get("SQ_INSTS_VALU", 0.0) / max(1, calls[k].get("SQ_INSTS_VALU", 1))
    lanes = acc[k].get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, acc[k].get("SQ_ACTIVE_INST_VALU", 1.0))
    out["kernels"][k] = {"valu_wave_instr_per_launch": round(valu), "lanes_active_per_valu_instr": round(lanes, 2), "launches_profiled": n, "fetch_size_kb_per_launch_raw": round(fetch_kb, 1), "write_size_kb_per_launch_raw": round(write_kb, 1),
                         "bytes_per_launch_corrected": round((2.0 * fetch_kb + write_kb) * 1024.0)}
e = out["kernels"].get("pt::k_extend")
out["extend_bytes_per_launch"] = e["bytes_per_launch_corrected"] if e else None
out["extend_valu_wave_instr_per_launch"] = e["valu_wave_instr_per_launch"] if e else None
out["extend_lanes_active_per_valu_instr"] = e["lanes_active_per_valu_instr"] if e else None
# vector-memory pipe of the extend kernel: TA / TD busy cycles (summed over the 256 CUs) against the dispatch's cycles
# (GRBM_GUI_ACTIVE is summed over the 8 XCDs), wave-level load instructions per launch
ke = "pt::k_extend" if "pt::k_extend" in acc else None   # not k_extend_packet (the coherent primaries): that has its own entry under "kernels"
if ke and acc[ke].get("GRBM_GUI_ACTIVE") and acc[ke].get("TA_TA_BUSY_sum"):
    cyc = acc[ke]["GRBM_GUI_ACTIVE"] / calls[ke]["GRBM_GUI_ACTIVE"] / 8.0
    out["extend_cycles_per_launch"] = round(cyc)
    out["extend_ta_busy_frac"] = round(acc[ke]["TA_TA_BUSY_sum"] / calls[ke]["TA_TA_BUSY_sum"] / 256.0 / cyc, 4)
    out["extend_td_busy_frac"] = round(acc[ke]["TD_TD_BUSY_sum"] / calls[ke]["TD_TD_BUSY_sum"] / 256.0 / cyc, 4)
    out["extend_ta_stalled_by_tc_frac"] = round(acc[ke]["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / calls[ke]["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / 256.0 / cyc, 4)
    out["extend_vmem_rd_wave_instr_per_launch"] = round(acc[ke]["SQ_INSTS_VMEM_RD"] / calls[ke]["SQ_INSTS_VMEM_RD"])
    out["extend_salu_instr_per_launch"] = round(acc[ke]["SQ_INSTS_SALU"] / calls[ke]["SQ_INSTS_SALU"])
    out["extend_l2_hit_rate"] = round(acc[ke]["TCC_HIT_sum"] / (acc[ke]["TCC_HIT_sum"] + acc[ke]["TCC_MISS_sum"]), 4)
for extra in sys.argv[7:]:          # optional key=value pairs: material_ext=1 russian_roulette=1 shared_triangle_phase=1
    k, v = extra.split("=")
    out[k] = int(v)
out["note"] = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes), mean over the k_extend launches of bench.py --steps 2; "
               "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 doubling of FETCH_SIZE applied although the BVH gathers are "
               "not the calibrated streaming pattern; fabric-side, Infinity-Cache hits included")
path = "profiles/%s/pmc_traffic.json" % os.environ.get("PT_PROFILE_ROUND", "r03")
os.makedirs(os.path.dirname(path), exist_ok=True)
cur = json.load(open(path)) if os.path.exists(path) else []
key = lambda c: (c["workload"], c["width"], c["height"], c["spp_per_step"], c["depth"], c.get("material_ext", 0), c.get("russian_roulette", 0),
                 c.get("shared_triangle_phase", 0))
cur = [c for c in cur if key(c) != key(out)] + [out]
json.dump(cur, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
