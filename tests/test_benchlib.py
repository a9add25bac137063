"""benchlib/ (the parts of bench.py outside the timed region) on fabricated profiler tables: the dominant kernel is chosen by summed
time per kernel name, the roofline object prices it against the right roof (MFMA above the machine balance, HBM below), the
16x16x32 kernels are priced against that shape's measured ceiling, and the traffic lookup reads the newest committed counter file.
No GPU."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import roofline  # noqa: E402
from benchlib.launch import CONFIGS, parse  # noqa: E402


def entry(kernel, kind, total_ms, launches, flops, n=32, d=32, h=128, w=128, cin=32, cout=32, k=(3, 3, 3), ups=0):
    shape = types.SimpleNamespace(n=n, d=d, h=h, w=w, cin=cin, cout=cout, kd=k[0], kh=k[1], kw=k[2], upsample_in=ups)
    return types.SimpleNamespace(kernel=kernel.encode(), kind=kind, total_ms=total_ms, launches=launches, flops_per_launch=flops,
                                 shape=shape)


def test_dominant_kernel_is_the_largest_summed_time_by_name():
    flops = 2.0 * 27 * 32 * 32 * 32 * 32 * 128 * 128
    table = [entry('conv_fwd5<bf16,2,2,3,3,3>', 0, 4.0, 10, 1e11, d=16, h=64, w=64, cin=64, cout=64),
             entry('conv_fwd3w<bf16,32->32>', 0, 3.5, 5, flops),
             entry('conv_fwd3w<bf16,32->32>', 0, 3.0, 2, 2 * flops, n=64)]
    name, dom, by = roofline.dominant(table)
    assert name == b'conv_fwd3w<bf16,32->32>' and dom is table[1]
    assert abs(by[name] - 6.5) < 1e-9


def test_roofline_object_mfma_bound_and_priced_against_the_shape_ceiling():
    flops = 2.0 * 27 * 32 * 32 * 32 * 32 * 128 * 128          # 32 -> 32, n32, 32 x 128 x 128
    table = [entry('conv_fwd3w<bf16,32->32>', 0, 3.5, 5, flops)]
    timed = [entry('conv_fwd3w<bf16,32->32>', 0, 0.72 * 50, 50, flops)]
    r = roofline.roofline_object(timed, table, b'conv_fwd3w<bf16,32->32>', 'bf16', False, 2)
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert abs(r['achieved'] - flops / 0.72e-3 / 1e12) < 0.5 and abs(r['frac'] - r['achieved'] / 2500.0) < 1e-3
    assert r['algorithmic_bytes'] == 32 * 32 * 128 * 128 * (32 + 32) * 2
    if os.path.exists(os.path.join(ROOT, 'profiles', 'r04_mfma_ceiling.txt')):
        # 16x16x32 kernels against that shape's ceiling (higher than the 32x32x16 one the other kernels are priced against)
        assert r['sustained_peak'] > roofline.sustained_mfma_peak('bf16', 'conv_fwd5<bf16,2,2,3,3,3>') > 1000.0
        assert abs(r['frac_of_sustained'] - r['achieved'] / r['sustained_peak']) < 1e-3
    assert r['by_shape'][0]['calls_per_step'] == 2.5
    # a replayed hipGraph has nothing to bracket: the object quotes the calibration table and says so
    rg = roofline.roofline_object([], table, b'conv_fwd3w<bf16,32->32>', 'bf16', True, 2)
    assert 'calibration' in rg['timing'] and abs(rg['avg_ms'] - 0.7) < 1e-6


def test_small_channel_layers_are_priced_in_bytes():
    # 2-D, 4 -> 8 channels at 1024^2, f32: 2 * 9 * 4 * 8 flops per pixel against 48 bytes per pixel -- far below the machine balance
    npx = 4 * 1024 * 1024
    e = entry('conv_small_fwd<f32>', 0, 0.1, 2, 2.0 * 9 * 4 * 8 * npx, n=4, d=1, h=1024, w=1024, cin=4, cout=8, k=(1, 3, 3))
    r = roofline.small_channel_object([e], 'f32')
    assert r['bound'] == 'hbm' and r['algorithmic_bytes'] == npx * 12 * 4
    assert abs(r['achieved'] - npx * 48 / 0.05e-3 / 1e9) < 1.0 and r['peak'] == 8000.0
    # the dominant-kernel object switches roofs by arithmetic intensity as well
    r2 = roofline.roofline_object([e], [e], b'conv_small_fwd<f32>', 'f32', False, 2)
    assert r2['bound'] == 'hbm'
    assert roofline.small_channel_object([entry('conv_fwd5<bf16,2,2,3,3,3>', 0, 1.0, 1, 1e12)], 'bf16') is None


def test_traffic_lookup_prefers_the_newest_counter_file():
    path = os.path.join(ROOT, 'profiles', 'r05_pmc_traffic.json')
    if not os.path.exists(path):
        return
    tab = json.load(open(path))
    hit = next(e for e in tab['entries'] if e['kind'] == 'fwd' and e.get('variant', '').startswith(('fwd bias', 'bias', 'with')))
    s = hit['shape']
    e = entry('x', 0, 1.0, 1, 1.0, n=s['n'], d=s['d'], h=s['h'], w=s['w'], cin=s['cin'], cout=s['cout'], k=tuple(s['k']))
    got = roofline.pmc_traffic(e, tab['dtype'])
    assert got is not None and got > 0
    assert roofline.pmc_traffic(e, 'f64') is None


def test_configs_follow_baseline_json(monkeypatch):
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert len(base['configs']) == 5 and set(CONFIGS) >= {1, 2, 3, 4, 5}
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = parse()
    assert (a.gpus, a.config, a.size, a.phase, a.batch, a.dtype, a.alpha) == (1, 3, 's', 6, 32, 'bf16', 0.0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', 'out_txt', '--batch', '4'])
    a = parse()
    assert (a.config, a.size, a.phase, a.batch) == (6, 'xs', 5, 4)
