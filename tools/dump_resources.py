"""Writes the per-kernel register / scratch / spill table of the current build to profiles/ (the copy the judge reads;
tests/test_build_resources.py checks that it matches the build)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from saragan_amd import build as b
    out = os.path.join(ROOT, 'profiles', sys.argv[1] if len(sys.argv) > 1 else 'r04_kernel_resources.json')
    u = b.resource_usage()
    with open(out, 'w') as f:
        json.dump(u, f, indent=1, sort_keys=True)
    n = sum(len(k) for k in u.values())
    worst = max(((v['vgprs'], k) for ks in u.values() for k, v in ks.items()))
    print(f'{out}: {n} kernels, max VGPRs {worst[0]} ({worst[1][:60]})')
