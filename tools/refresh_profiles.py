"""Copies what tools/final_collect1.sh and tools/final_collect2.sh left under gpurun_out/final/ into profiles/r02_*
(the judged, committed copies) and derives the two summaries that join counters with the probe's manifest.
usage: python tools/refresh_profiles.py          (from the repo root, after both collection calls)"""
import csv
import gzip
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'final')
DST = os.path.join(ROOT, 'profiles')
TAG = os.environ.get('PROFILE_TAG', 'r03_')


def copy(src, dst, gz=False):
    s = os.path.join(SRC, src)
    if not os.path.exists(s):
        print('missing', src)
        return False
    d = os.path.join(DST, TAG + dst)
    if gz:
        with open(s, 'rb') as f, gzip.open(d + '.gz', 'wb') as g:
            shutil.copyfileobj(f, g)
    else:
        shutil.copyfile(s, d)
    return True


def last_json_line(path):
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith('{')]
    return lines[-1] if lines else None


def strip_counter_csv(src, dst):
    """Counter CSVs keep our kernels only (torch's template names are kilobytes per row)."""
    s = os.path.join(SRC, src)
    if not os.path.exists(s):
        print('missing', src)
        return
    rows = list(csv.reader(open(s)))
    head, body = rows[0], rows[1:]
    ki = head.index('Kernel_Name')
    keep = [r for r in body if 'at::native' not in r[ki] and 'rocclr' not in r[ki]]
    with open(os.path.join(DST, TAG + dst), 'w', newline='') as f:
        csv.writer(f).writerows([head] + keep)


def sq_wait_summary():
    path = os.path.join(SRC, 'pmc_sq', 'sq_counter_collection.csv')
    if not os.path.exists(path):
        print('missing pmc_sq')
        return
    per = {}
    order = []
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        if 'conv_' not in k:
            continue
        did = int(r['Dispatch_Id'])
        key = k.split('(')[0]
        if key not in per:
            per[key] = dict(dispatch=did, grid=int(r['Grid_Size']), c={})
            order.append(key)
        if per[key]['dispatch'] == did:
            per[key]['c'][r['Counter_Name']] = float(r['Counter_Value'])
    entries = []
    for key in order:
        c = per[key]['c']
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        if wc <= 0:
            continue
        busy = c.get('SQ_BUSY_CYCLES', 0.0)
        entries.append(dict(kernel=key, grid=per[key]['grid'], wave_quad_cycles=wc,
                            wait_any_frac=round(c.get('SQ_WAIT_ANY', 0.0) / wc, 3),
                            wait_inst_any_frac=round(c.get('SQ_WAIT_INST_ANY', 0.0) / wc, 3),
                            wait_inst_lds_frac=round(c.get('SQ_WAIT_INST_LDS', 0.0) / wc, 3),
                            active_inst_frac=round(c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 3),
                            lds_bank_conflict_cycles=c.get('SQ_LDS_BANK_CONFLICT', 0.0),
                            lds_idx_active_cycles=c.get('SQ_LDS_IDX_ACTIVE', 0.0),
                            sq_busy_cycles=busy))
    note = ('rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY '
            'SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE over tools/pmc_probe.py (raw: r02_pmc_SQ_WAIT.csv.gz); fractions are of '
            'SQ_WAVE_CYCLES (summed over waves); first launch of each kernel instantiation')
    json.dump(dict(note=note, entries=entries), open(os.path.join(DST, TAG + 'pmc_sq_wait.json'), 'w'), indent=0)


def main():
    os.makedirs(DST, exist_ok=True)
    line = last_json_line(os.path.join(SRC, 'bench_default.json'))
    if line:
        open(os.path.join(DST, TAG + 'bench_line.json'), 'w').write(line + '\n')
    copy('bench_conv_table.txt', 'bench_conv_table.txt')
    others = [last_json_line(os.path.join(SRC, f'bench_config{c}.json')) for c in (1, 2, 4, 5)
              if os.path.exists(os.path.join(SRC, f'bench_config{c}.json'))]
    if os.path.exists(os.path.join(SRC, 'bench_mixing.json')):      # configs[2] in a mixing phase (alpha 0.5, freeze train ops)
        others.append(last_json_line(os.path.join(SRC, 'bench_mixing.json')))
    open(os.path.join(DST, TAG + 'bench_other_configs.jsonl'), 'w').write('\n'.join(o for o in others if o) + '\n')
    copy('bench_reference_point.jsonl', 'bench_reference_point.jsonl')
    copy('bench_repeats.jsonl', 'bench_repeats.jsonl')
    log = os.path.join(SRC, 'pytest_gpu.log')
    if os.path.exists(log):
        open(os.path.join(DST, TAG + 'pytest_gpu_summary.txt'), 'w').write('\n'.join(open(log).read().splitlines()[-4:]) + '\n')
    copy(os.path.join('prof_bench', 'bench_kernel_stats.csv'), 'bench_kernel_stats.csv')
    copy(os.path.join('prof_bench', 'bench_domain_stats.csv'), 'bench_domain_stats.csv')
    for f in ('elementwise_roofline.json', 'elementwise_roofline.txt', 'store_war_probe.txt', 'mfma_lds_probe.txt',
              'pk_f32_probe.txt', 'phase_stamps.txt', 'pmc_manifest.json'):
        copy(f, f)
    strip_counter_csv(os.path.join('pmc_rd', 'rd_counter_collection.csv'), 'pmc_FETCH_SIZE.csv')
    strip_counter_csv(os.path.join('pmc_wr', 'wr_counter_collection.csv'), 'pmc_WRITE_SIZE.csv')
    copy(os.path.join('pmc_mfma', 'mf_counter_collection.csv'), 'pmc_MFMA_BUSY.csv', gz=True)
    copy(os.path.join('pmc_sq', 'sq_counter_collection.csv'), 'pmc_SQ_WAIT.csv', gz=True)
    out = os.path.join(DST, TAG + 'pmc_traffic.json')
    rc = subprocess.call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), os.path.join(SRC, 'pmc_manifest.json'),
                          os.path.join(SRC, 'pmc_rd', 'rd_counter_collection.csv'),
                          os.path.join(SRC, 'pmc_wr', 'wr_counter_collection.csv'),
                          os.path.join(SRC, 'pmc_mfma', 'mf_counter_collection.csv'), out],
                         stdout=open(os.path.join(DST, TAG + 'pmc_traffic.txt'), 'w'))
    print('pmc_summary rc', rc)
    sq_wait_summary()
    line = last_json_line(os.path.join(SRC, 'bench_gloo2.json')) if os.path.exists(os.path.join(SRC, 'bench_gloo2.json')) else None
    if line:      # the launcher's own smoke: two ranks stacked on the one GPU of the box, gloo collectives (NOT a scaling number)
        open(os.path.join(DST, TAG + 'bench_gloo_rehearsal.json'), 'w').write(line + '\n')
    for rep in ('loss_curve_report_wgan.json', 'loss_curve_report_logistic_mix.json', 'loss_curve_report_cfg3.json'):
        s = os.path.join(ROOT, 'gpurun_out', rep)
        if os.path.exists(s):
            shutil.copyfile(s, os.path.join(DST, TAG + rep))


if __name__ == '__main__':
    main()
