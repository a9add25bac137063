"""Per-kernel mean durations of the timed regions of ONE bench.py process, side by side, from a rocprofv3 --kernel-trace
CSV (the legs that run after the main loop read 54-57 ms/step on some boxes against 65+ in the main
loop).  bench.py, run with SARAGAN_BENCH_MARK=1, launches a marker kernel (a 3-element torch cumsum) at the begin and
the end of every timed region; consecutive marker pairs delimit the windows (main loop, loader leg, faded-branch leg,
fp32 leg).  Also reports, per window, the summed kernel time, the span from first kernel start to last kernel end and
the idle share in between (launch gaps).

usage: python tools/trace_windows.py <kernel_trace.csv> [marker-regex] > profiles/r03_leg_windows.txt
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<[^(]{0,60})?)', name)
    return (m.group(1) if m else name)[:70]


def main():
    path = sys.argv[1]
    marker = re.compile(sys.argv[2] if len(sys.argv) > 2 else r'cumsum|scan|Scan')
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker.search(r[2])]
    print(f'{len(rows)} kernel dispatches, {len(marks)} marker launches')
    windows = [(marks[i], marks[i + 1]) for i in range(0, len(marks) - 1, 2)]
    names = ['main loop', 'loader leg', 'faded-branch leg', 'fp32 leg'] + [f'window {i}' for i in range(4, 20)]
    per = []
    for wi, (a, b) in enumerate(windows):
        sel = rows[a + 1:b]
        if not sel:
            continue
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in sel:
            k = short(n)
            agg[k][0] += 1
            agg[k][1] += e - s
        busy = sum(e - s for s, e, _ in sel)
        span = max(e for _, e, _ in sel) - sel[0][0]
        # (kernels of one stream do not overlap; a side stream's copies are not kernel dispatches)
        print(f'[{wi}] {names[wi]}: {len(sel)} dispatches, kernel time {busy / 1e6:.2f} ms, span {span / 1e6:.2f} ms, '
              f'idle between kernels {100.0 * (span - busy) / span:.1f} %')
        per.append((names[wi], agg, busy, span))
    if len(per) < 2:
        return
    base = per[0][1]
    keys = sorted(base, key=lambda k: -base[k][1])[:40]
    hdr = f"{'kernel':70s} " + ' '.join(f'{n[:16]:>16s}' for n, *_ in per) + '   (mean us per launch; launches per window in brackets on the first)'
    print(hdr)
    for k in keys:
        cells = []
        for _, agg, _, _ in per:
            c, t = agg.get(k, (0, 0))
            cells.append(f'{t / c / 1e3:16.1f}' if c else f"{'-':>16s}")
        print(f'{k:70s} ' + ' '.join(cells) + f'   [{base[k][0]}]')
    print('ratio of summed kernel time per launch-weighted common kernels, window / main loop:')
    for n, agg, _, _ in per[1:]:
        num = den = 0.0
        for k, (c, t) in agg.items():
            if k in base and c and base[k][0]:
                num += t / c * base[k][0]
                den += base[k][1]
        if den:
            print(f'  {n}: {num / den:.3f}')


if __name__ == '__main__':
    main()
