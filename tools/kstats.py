"""Prints a rocprofv3 --stats kernel table per step: python tools/kstats.py <kernel_stats.csv> <steps run under the profiler> [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total {tot / 1e6 / steps:.3f} ms/step over {steps:.0f} steps, {sum(int(r["Calls"]) for r in rows) / steps:.0f} launches/step')
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.2f}")
