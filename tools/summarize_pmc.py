"""Aggregates rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/pass*/**/*counter_collection.csv) per kernel."""
import csv
import glob
import sys
from collections import defaultdict

tag = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(f"gpurun_out/pmc_{tag}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    c = acc[k]
    n = max(calls[k].values())
    print(f"== {k}  dispatches={n}")
    for name in sorted(c):
        print(f"   {name:34s} {c[name]:18.0f}   per-dispatch {c[name] / max(1, calls[k][name]):16.1f}")
    if c.get("SQ_INSTS_VALU"):
        print(f"   -> lanes active per VALU inst: {c['SQ_THREAD_CYCLES_VALU'] / max(1.0, c['SQ_ACTIVE_INST_VALU']) :.2f} (of 64? see units)  "
              f"thread_cycles/insts={c['SQ_THREAD_CYCLES_VALU'] / c['SQ_INSTS_VALU']:.2f}")
    if c.get("SQ_WAVE_CYCLES"):
        print(f"   -> wait_any/wave_cycles={c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}  wait_inst_any={c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}  "
              f"valu_active={c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_WAVE_CYCLES']:.3f}")
    if c.get("TCC_HIT_sum"):
        print(f"   -> L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4f}")
    if c.get("FETCH_SIZE"):
        print(f"   -> FETCH_SIZE KB/dispatch {c['FETCH_SIZE'] / max(1, calls[k]['FETCH_SIZE']):.1f} (x2 for wide streaming reads on gfx950), WRITE_SIZE KB/dispatch {c.get('WRITE_SIZE', 0) / max(1, calls[k].get('WRITE_SIZE', 1)):.1f}")
