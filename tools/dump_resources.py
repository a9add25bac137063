"""Writes the per-kernel register / scratch / spill table of the current build to profiles/ (the copy the judge reads;
tests/test_build_resources.py checks that it matches the build)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from saragan_amd import build as b
    out = os.path.join(ROOT, 'profiles', sys.argv[1] if len(sys.argv) > 1 else 'r05_kernel_resources.json')
    u = b.resource_usage()
    with open(out, 'w') as f:
        json.dump(u, f, indent=1, sort_keys=True)
    n = sum(len(k) for k in u.values())
    worst = max(((v['vgprs'], k) for ks in u.values() for k, v in ks.items()))
    print(f'{out}: {n} kernels, max VGPRs {worst[0]} ({worst[1][:60]})')
    # where the spilled SGPRs are re-read: inside or outside the unrolled MFMA loops (tests/test_build_resources.py)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_build_resources import sgpr_spill_sites
    rows = []
    for src in ('conv3d.hip', 'conv3p.hip', 'conv3w.hip', 'wgrad.hip'):
        with open(b.device_asm(src)) as f:
            for k, (i_, o_) in sorted(sgpr_spill_sites(f).items()):
                spill = u[src].get(k, {}).get('sgpr_spill', '?')
                rows.append(f'{src:11s} sgpr_spill {spill!s:>4}  v_readlane/v_writelane inside K loops {i_:4d}  outside {o_:5d}  {k}')
    with open(os.path.join(ROOT, 'profiles', 'r05_sgpr_spill_sites.txt'), 'w') as f:
        f.write('SGPR spill traffic (v_writelane_b32 / v_readlane_b32) of the MFMA kernels by site, from the kept device assembly\n')
        f.write('\n'.join(rows) + '\n')
    print(f'profiles/r05_sgpr_spill_sites.txt: {len(rows)} kernels')
