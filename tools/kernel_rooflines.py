"""Per-kernel roofline table of one workload from committed files: fabric-side bytes and VALU wave-instructions per launch
(profiles/<round>/pmc_<workload>_summary.txt: separate rocprofv3 --pmc passes, tools/summarize_pmc.py) over the kernel's average
launch time (rocprofv3 --kernel-trace --stats of the same build: bench_atrium_kernel_stats.csv / kernel_stats_<workload>.txt).
  bytes  = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 doubling of FETCH_SIZE as in tools/pmc_traffic.py)  vs  8 TB/s
  valu   = SQ_INSTS_VALU  vs  1228.8 G wave-instructions/s (256 CUs x 4 SIMD x 2.4 GHz / 2 cycles: every instruction at full rate)
  lanes  = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / (cycles per instruction, = 1 in this counter's units): active lanes per VALU instruction
usage: python tools/kernel_rooflines.py <round> <workload>     e.g. r03 atrium
"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK, VALU_PEAK = 8.0e12, 1228.8e9


def pmc_blocks(path):
    out = {}
    for b in re.split(r"^== ", open(path).read(), flags=re.M)[1:]:
        head = b.splitlines()[0]
        name, nd = head.split("dispatches=")[0].strip(), int(head.split("dispatches=")[1])
        d = {}
        for ln in b.splitlines()[1:]:
            m = re.match(r"\s+(\S+)\s+(\d+)\s+per-dispatch\s+([\d.]+)", ln)
            if m:
                d[m.group(1)] = float(m.group(3))
        out[name] = (nd, d)
    return out


def durations(rnd, workload):
    """kernel name -> (calls, average ms)"""
    out = {}
    if workload == "atrium":
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", rnd, "bench_atrium_kernel_stats.csv"))):
            out[r["Name"].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    else:
        for ln in open(os.path.join(ROOT, "profiles", rnd, f"kernel_stats_{workload}.txt")):
            m = re.match(r"(.*?)\s+calls=\s*(\d+)\s+total_ms=\s*([\d.]+)\s+avg_us=\s*([\d.]+)", ln)
            if m:
                out[m.group(1).replace("void ", "")] = (int(m.group(2)), float(m.group(4)) / 1e3)
    return out


def main():
    rnd, workload = sys.argv[1], sys.argv[2]
    pmc = pmc_blocks(os.path.join(ROOT, "profiles", rnd, f"pmc_{workload}_summary.txt"))
    dur = durations(rnd, workload)
    print(f"# {workload}, profiles/{rnd}: per launch, counters from the --pmc passes, time from the kernel trace of the same build")
    print(f"{'kernel':34s} {'ms':>8s} {'GB':>8s} {'TB/s':>6s} {'of HBM':>7s} {'Ginstr':>8s} {'G/s':>7s} {'of VALU':>8s} {'lanes':>6s}  bound")
    for name, (nd, d) in pmc.items():
        if not name.startswith("pt::k_") or "k_primary_setup" in name or "k_transmittance" in name or "k_set_count" in name:
            continue
        key = name.replace("void ", "")
        cands = [(k, v) for k, v in dur.items() if k.startswith(key) or key.startswith(k[:len(key)])]
        if not cands:
            continue
        # several instantiations can share the truncated name (timing / counting variants): the one with most calls is the shipped one
        k, (calls, ms) = max(cands, key=lambda kv: kv[1][0])
        by = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
        vi = d.get("SQ_INSTS_VALU", 0.0)
        lanes = d.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(vi, 1.0)
        fb, fv = by / (ms * 1e-3) / HBM_PEAK, vi / (ms * 1e-3) / VALU_PEAK
        print(f"{name[:34]:34s} {ms:8.3f} {by / 1e9:8.2f} {by / (ms * 1e-3) / 1e12:6.2f} {fb:7.3f} {vi / 1e9:8.3f} {vi / (ms * 1e-3) / 1e9:7.1f} {fv:8.3f} {lanes:6.1f}  "
              f"{'hbm' if fb >= fv else 'valu'}")


if __name__ == "__main__":
    main()
