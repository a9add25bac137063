"""Samples the GPU's power and shader clock from sysfs (hwmon power1_average / power1_input, freq1_input; pp_dpm_sclk's
active line) every ~20 ms while `python bench.py` runs with leg markers, and prints per-leg averages next to the raw
samples: does a leg that runs its MFMA kernels 20 % faster (profiles/r03_leg_windows.txt) draw the same power at a higher
clock?   usage: python tools/sysfs_trace.py [bench args...] > profiles/r03_sysfs_trace.txt"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find():
    out = {}
    for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        for name in ('power1_average', 'power1_input', 'freq1_input', 'temp2_input', 'power1_cap'):
            p = os.path.join(hw, name)
            if os.path.exists(p) and name not in out:
                out[name] = p
    return out


def read(p):
    try:
        with open(p) as f:
            return int(f.read().strip())
    except Exception:
        return -1


def main():
    files = find()
    print('sysfs files:', files, flush=True)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.time(), read(files.get('power1_average', files.get('power1_input', ''))),
                            read(files.get('freq1_input', '')), read(files.get('temp2_input', ''))))
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    env = dict(os.environ, SARAGAN_BENCH_MARK='1')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline'] + sys.argv[1:], env=env,
                       capture_output=True, text=True, timeout=600)
    stop.set()
    th.join()
    marks = []
    for ln in p.stdout.splitlines():
        if ln.startswith('MARK '):
            _, t, what = ln.split(' ', 2)
            marks.append((float(t), what))
        elif ln.startswith('{'):
            print('bench line:', ln[:600], '...')
    print('cap (uW):', read(files.get('power1_cap', '')))
    begins = [t for t, w in marks if w.startswith('timed region begins')]
    ends = [(t, w) for t, w in marks if w.startswith('timed region ends')]
    names = ['main loop', 'loader leg', 'faded-branch leg', 'fp32 leg']
    for i, (b, (e, w)) in enumerate(zip(begins, ends)):
        sel = [s for s in samples if b <= s[0] <= e]
        if sel:
            pw = [s[1] for s in sel if s[1] >= 0]
            fq = [s[2] for s in sel if s[2] >= 0]
            print(f'{names[i] if i < 4 else i}: {w}; {len(sel)} samples, power mean {sum(pw) / max(1, len(pw)) / 1e6:.0f} W '
                  f'max {max(pw or [0]) / 1e6:.0f} W, sclk mean {sum(fq) / max(1, len(fq)) / 1e6:.0f} MHz')
    t0 = samples[0][0] if samples else 0
    print('raw samples (s since start, W, MHz, C), every 5th:')
    for s in samples[::5]:
        print(f'{s[0] - t0:8.2f} {s[1] / 1e6:8.0f} {s[2] / 1e6:8.0f} {s[3] / 1e3:6.1f}')
    if p.returncode != 0:
        print(p.stderr[-2000:])


if __name__ == '__main__':
    main()
