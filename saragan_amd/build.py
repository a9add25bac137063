"""Builds libsaragan_hip.so for gfx950 with hipcc (in-tree: saragan_amd/libsaragan_hip.so)."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsaragan_hip.so')
SOURCES = ['conv3d.hip', 'conv3p.hip', 'conv3w.hip', 'wgrad.hip', 'elementwise.hip', 'optim.hip', 'prof.hip', 'small.hip', 'subpix.hip', 'gemm.hip', 'metrics.hip']
HEADERS = ['common.h', 'prof.h', 'conv_args.h', os.path.join('..', '..', 'include', 'saragan_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# The MFMA kernels' off-phases share their SIMD with the other wave group's MFMAs, and packed-f32 VALU ops
# (v_pk_mul_f32 / v_pk_add_f32, which the SLP vectoriser forms from the epilogues' per-element arithmetic) were measured at
# ~110 cycles EACH there -- they queue behind the matrix pipe -- against 4-8 for the scalar forms: a 32-element masked
# epilogue took 4.3k cycles instead of 0.7k (tools/ts_conv.py, SG_DBG_FLAGS=128|256).  No SLP in those two files.
# -save-temps=obj: the device assembly of exactly the code that ships (build/<name>-hip-amdgcn-amd-amdhsa-gfx950.s) is kept
# for tests/test_build_resources.py, which checks the hand-counted `s_waitcnt lgkmcnt(N)` regions of the unrolled K loops.
FILE_FLAGS = {'conv3d.hip': ['-fno-slp-vectorize', '-save-temps=obj'], 'wgrad.hip': ['-fno-slp-vectorize', '-save-temps=obj'],
              'subpix.hip': ['-fno-slp-vectorize'], 'conv3p.hip': ['-fno-slp-vectorize', '-save-temps=obj'],
              'conv3w.hip': ['-fno-slp-vectorize', '-save-temps=obj'], 'gemm.hip': ['-fno-slp-vectorize']}


def _hipcc():
    c = os.environ.get('HIPCC')
    if c:
        return c
    return '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc'


def device_asm(source):
    """Path of the gfx950 assembly hipcc kept for `source` (conv3d.hip / wgrad.hip), building if it is not there."""
    path = os.path.join(HERE, 'build', source.replace('.hip', '') + '-hip-amdgcn-amd-amdhsa-gfx950.s')
    if not os.path.exists(path) or needs_build():
        build(force=True, verbose=False)
    return path


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


RESOURCES = os.path.join(HERE, 'build', 'resource_usage.json')
_REMARK = re.compile(r'remark:\s+(?:\S+:\d+:\d+:\s+)?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                     r'SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]):\s+(\S+)')
_KEYS = {'TotalSGPRs': 'sgprs', 'VGPRs': 'vgprs', 'AGPRs': 'agprs', 'ScratchSize [bytes/lane]': 'scratch_bytes',
         'Occupancy [waves/SIMD]': 'occupancy', 'SGPRs Spill': 'sgpr_spill', 'VGPRs Spill': 'vgpr_spill',
         'LDS Size [bytes/block]': 'static_lds'}


def parse_resource_remarks(text):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgprs, vgpr_spill, ...}}."""
    out, cur = {}, None
    for m in _REMARK.finditer(text):
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[_KEYS[k]] = int(v)
    return out


def build(force=False, verbose=True):
    if not force and not needs_build() and os.path.exists(RESOURCES):
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, 'build', s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(s, []) + ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        log = open(o + '.log', 'w+')
        procs.append((s, subprocess.Popen(cmd, stderr=log), log))
    usage = {}
    for s, p, log in procs:
        rc = p.wait()
        log.seek(0)
        text = log.read()
        log.close()
        if rc != 0:
            sys.stderr.write(text)
            raise RuntimeError(f'hipcc failed on {s}')
        # the remarks go to the per-kernel table; anything else the compiler said (warnings) is shown
        rest = [ln for ln in text.splitlines() if 'kernel-resource-usage' not in ln]
        rest = [ln for i, ln in enumerate(rest) if not _is_remark_context(rest, i)]
        if verbose and any('warning' in ln or 'error' in ln for ln in rest):
            sys.stderr.write('\n'.join(rest) + '\n')
        usage[s] = parse_resource_remarks(text)
    for f_ in os.listdir(os.path.join(HERE, 'build')):      # of the saved temporaries only the device assembly is kept
        if f_.endswith(('.bc', '.hipi', '.out', '.hipfb', '.resolution.txt', '-unknown-linux-gnu.s')) or \
                f_.endswith('-gfx950.o'):
            os.unlink(os.path.join(HERE, 'build', f_))
    with open(RESOURCES, 'w') as f:
        json.dump(usage, f, indent=1, sort_keys=True)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def _is_remark_context(lines, i):
    """clang prints the source line and a caret under every remark: drop those too."""
    s = lines[i].lstrip()
    return bool(re.match(r'^\d+ \|', s)) or s.startswith('| ^') or s == '|' or s.startswith('^')


def resource_usage():
    """Per-kernel register / scratch / spill table of the current build (built on demand)."""
    build(force=not os.path.exists(RESOURCES), verbose=False)
    with open(RESOURCES) as f:
        return json.load(f)


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
