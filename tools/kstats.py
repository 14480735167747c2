"""Prints a rocprofv3 --stats kernel_stats.csv compactly: python tools/kstats.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'].replace('void ', '')[:40]:40s} calls={r['Calls']:>4s} total_ms={int(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={float(r['Percentage']):6.2f}")
