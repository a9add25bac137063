"""Joins tools/pmc_probe.py's manifest with the rocprofv3 counter CSVs of its three passes.

usage: python tools/pmc_summary.py gpurun_out/pmc_manifest.json gpurun_out/pmc_rd/rd_counter_collection.csv \
           gpurun_out/pmc_wr/wr_counter_collection.csv gpurun_out/pmc_mfma/mf_counter_collection.csv profiles/r02_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB and reports half of the bytes of wide coalesced reads on
gfx950 -> doubled; WRITE_SIZE (KiB) is exact.  The calibration launch of the probe (sg_axpby over 2^28 bf16: 1 GiB read,
0.5 GiB written) is checked against both before anything else is trusted.  (What FETCH_SIZE reports for 32-byte sliver
reads is calibrated separately: tools/probe/sliver_probe.hip.)  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel
cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; for v_mfma_f32_32x32x16_bf16 the busy cycles are 32 per instruction,
which the summary cross-checks against FLOPs / 32768 per instruction."""
import csv
import json
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


def per_dispatch(rows, counter):
    out = {}
    for r in rows:
        if r['Counter_Name'] == counter:
            out[int(r['Dispatch_Id'])] = (r['Kernel_Name'], float(r['Counter_Value']))
    return [(k, *out[k]) for k in sorted(out)]


def match(manifest, disp):
    """For every manifest entry the LAST call's matching dispatches (first-touch effects gone), in order: one dispatch,
    or the sum over `dispatches_per_call` of them (the two passes of a layer split over its input channels)."""
    res, pos = [], 0
    for m in manifest:
        found = []
        while pos < len(disp) and len(found) < m['repeats']:
            if m['match'] in disp[pos][1]:
                found.append(disp[pos])
            pos += 1
        assert len(found) == m['repeats'], (m['tag'], len(found))
        last = found[-m.get('dispatches_per_call', 1):]
        name = last[-1][1] if len(last) == 1 else ' + '.join(d[1].split('(')[0][:40] for d in last)
        res.append((last[-1][0], name, sum(d[2] for d in last)))
    return res


def main():
    man_p, rd_p, wr_p, mf_p, out_p = sys.argv[1:6]
    manifest = json.load(open(man_p))
    R = match(manifest, per_dispatch(load(rd_p), 'FETCH_SIZE'))
    W = match(manifest, per_dispatch(load(wr_p), 'WRITE_SIZE'))
    mf = load(mf_p)
    MB = match(manifest, per_dispatch(mf, 'SQ_VALU_MFMA_BUSY_CYCLES'))
    GA = match(manifest, per_dispatch(mf, 'GRBM_GUI_ACTIVE'))
    cal_r, cal_w = 2 * R[0][2] / (1 << 20), W[0][2] / (1 << 20)
    assert abs(cal_r - 1.0) < 0.02 and abs(cal_w - 0.5) < 0.02, ('calibration', cal_r, cal_w)
    res = {'unit': 'bytes per launch', 'dtype': 'bf16',
           'correction': 'FETCH_SIZE KiB x 2 (gfx950 wide reads), WRITE_SIZE KiB x 1',
           'calibration': {'axpby_read_GiB': cal_r, 'axpby_write_GiB': cal_w}, 'entries': []}
    for m, r, w, b, g in zip(manifest[1:], R[1:], W[1:], MB[1:], GA[1:]):
        fr, fw = 2.0 * r[2] * 1024, w[2] * 1024
        e = {'tag': m['tag'], 'kernel': r[1].replace('(anonymous namespace)::', '').split('(')[0][:70], 'read_bytes': fr, 'write_bytes': fw, 'traffic_bytes': fr + fw,
             'algorithmic_bytes': m['algorithmic_bytes'], 'traffic_over_algorithmic': (fr + fw) / m['algorithmic_bytes']}
        kind = 'fwd' if m['tag'].startswith('fwd') else ('wgrad' if m['tag'].startswith('wgrad') else 'elementwise')
        e['kind'] = kind
        if kind != 'elementwise':
            import re
            mm = re.search(r'n(\d+) (\d+)x(\d+)x(\d+) (\d+)->(\d+)', m['tag'])
            n, d, h, w_, ci, co = (int(v) for v in mm.groups())
            e['shape'] = {'n': n, 'd': d, 'h': h, 'w': w_, 'cin': ci, 'cout': co, 'k': [3, 3, 3]}
            e['variant'] = m['tag'].split(' n')[0]
            cycles = g[2] / 8.0
            e['kernel_cycles'] = cycles
            e['mfma_busy_cycles'] = b[2]
            e['mfma_busy_frac'] = b[2] / (1024.0 * cycles) if cycles else None
            e['mfma_busy_expected_from_flops'] = m['flops'] / 32768.0 * 32.0
        res['entries'].append(e)
    json.dump(res, open(out_p, 'w'), indent=1)
    for e in res['entries']:
        extra = f" mfma-busy {e['mfma_busy_frac']:.3f}" if e.get('mfma_busy_frac') is not None else ''
        print(f"{e['tag']:58s} read {e['read_bytes'] / 1e6:9.1f} MB write {e['write_bytes'] / 1e6:9.1f} MB alg {e['algorithmic_bytes'] / 1e6:9.1f} MB "
              f"x{e['traffic_over_algorithmic']:.2f}{extra}")


if __name__ == '__main__':
    main()
