"""Turns the two rocprofv3 counter passes of tools/pmc_probe.py into profiles/<round>_pmc_traffic.json.

usage: python tools/pmc_summary.py gpurun_out/pmc_rd/rd_counter_collection.csv gpurun_out/pmc_wr/wr_counter_collection.csv \
           profiles/r01_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB and reports half of the bytes of wide coalesced reads on
gfx950 -> doubled; WRITE_SIZE (KiB) is exact.  The calibration kernel in the probe (sg_axpby over 2^28 bf16: 1 GiB
read, 0.5 GiB written) is checked against both before anything else is trusted."""
import csv
import json
import sys

from pmc_probe import SHAPES


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


def main():
    rd, wr, out = sys.argv[1:4]
    R, W = load(rd, 'FETCH_SIZE'), load(wr, 'WRITE_SIZE')
    cal_r = [float(r['Counter_Value']) for r in R if 'axpby' in r['Kernel_Name']]
    cal_w = [float(r['Counter_Value']) for r in W if 'axpby' in r['Kernel_Name']]
    assert cal_r and abs(2 * cal_r[-1] / (1 << 20) - 1.0) < 0.02, ('FETCH_SIZE calibration', cal_r)
    assert cal_w and abs(cal_w[-1] / (1 << 19) - 1.0) < 0.02, ('WRITE_SIZE calibration', cal_w)

    def convs(rows):
        return [r for r in rows if any(k in r['Kernel_Name'] for k in ('conv_fwd', 'conv_wgrad'))]
    cr, cw = convs(R), convs(W)
    assert len(cr) == len(cw) == 6 * len(SHAPES), (len(cr), len(cw))
    res = {'unit': 'bytes per launch', 'dtype': 'bf16',
           'correction': 'FETCH_SIZE KiB x 2 (gfx950 wide reads), WRITE_SIZE KiB x 1',
           'calibration': {'axpby_read_GiB': 2 * cal_r[-1] / (1 << 20), 'axpby_write_GiB': cal_w[-1] / (1 << 20)},
           'entries': []}
    for i, (n, (d, h, w), cin, cout) in enumerate(SHAPES):
        blk_r, blk_w = cr[6 * i:6 * i + 6], cw[6 * i:6 * i + 6]
        vox = n * d * h * w
        for j, (kind, variant) in enumerate((('fwd', 'bias+lrelu+sign_out'), ('fwd', 'mask_bits'), ('wgrad', 'with dbias'))):
            # second repetition of each (index 3 + j): first touch effects gone
            fr = 2.0 * float(blk_r[3 + j]['Counter_Value']) * 1024
            fw = float(blk_w[3 + j]['Counter_Value']) * 1024
            if kind == 'fwd':
                alg = vox * (cin + cout) * 2 + 27 * cin * cout * 2
            else:
                alg = vox * (cin + cout) * 2 + 27 * cin * cout * 4
            res['entries'].append({'kind': kind, 'variant': variant, 'kernel': blk_r[3 + j]['Kernel_Name'].split('(')[0][:60],
                                   'shape': {'n': n, 'd': d, 'h': h, 'w': w, 'cin': cin, 'cout': cout, 'k': [3, 3, 3]},
                                   'read_bytes': fr, 'write_bytes': fw, 'traffic_bytes': fr + fw,
                                   'algorithmic_bytes': alg, 'traffic_over_algorithmic': (fr + fw) / alg})
    json.dump(res, open(out, 'w'), indent=1)
    for e in res['entries']:
        s = e['shape']
        print(f"{e['kind']:5s} {e['variant']:20s} {s['d']}x{s['h']}x{s['w']} {s['cin']:3d}->{s['cout']:3d} read {e['read_bytes'] / 1e6:8.1f} MB "
              f"write {e['write_bytes'] / 1e6:8.1f} MB  alg {e['algorithmic_bytes'] / 1e6:8.1f} MB  x{e['traffic_over_algorithmic']:.2f}")


if __name__ == '__main__':
    main()
