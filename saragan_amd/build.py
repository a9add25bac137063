"""Builds libsaragan_hip.so for gfx950 with hipcc (in-tree: saragan_amd/libsaragan_hip.so)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsaragan_hip.so')
SOURCES = ['conv3d.hip', 'wgrad.hip', 'elementwise.hip', 'optim.hip', 'prof.hip']
HEADERS = ['common.h', 'prof.h', os.path.join('..', '..', 'include', 'saragan_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    c = os.environ.get('HIPCC')
    if c:
        return c
    return '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [_hipcc()] + FLAGS + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
