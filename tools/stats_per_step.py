"""Per-step kernel table from a rocprofv3 --kernel-trace --stats run of bench.py: calls per step are derived from a kernel known
to run once per step (adam_ema_kernel: two launches per step).  usage: python tools/stats_per_step.py <kernel_stats.csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
adam = [r for r in rows if 'adam_ema_kernel' in r['Name']]
steps = int(adam[0]['Calls']) / 2 if adam else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'steps {steps:.0f}   kernel ms/step {tot / 1e6 / steps:.3f}   launches/step {sum(int(r["Calls"]) for r in rows) / steps:.1f}')
for r in rows[:top]:
    print(f"{r['Name'][:86]:86s} {int(r['Calls']) / steps:7.1f} {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms {float(r['AverageNs']) / 1e3:9.1f} us")
small = [r for r in rows if float(r['AverageNs']) < 12e3]
print(f'kernels under 12 us: {sum(int(r["Calls"]) for r in small) / steps:.1f} launches/step, {sum(float(r["TotalDurationNs"]) for r in small) / 1e6 / steps:.3f} ms/step')
