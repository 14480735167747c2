#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 --kernel-trace CSV: start/end (ms, relative) and name of every dispatch of the
last pt_render, to see what overlaps what. usage: tools/trace_timeline.py <dir> [n_last]"""
import csv, glob, sys
f = sorted(glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:9.3f} {e:9.3f} {e - s:8.3f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:40]}")
