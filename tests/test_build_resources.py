"""CPU test of the gfx950 code objects themselves: no kernel of libsaragan_hip.so may spill vector registers or use
scratch memory (hipcc's -Rpass-analysis=kernel-resource-usage remarks, captured by saragan_amd/build.py at build
time).  Round 1 shipped conv_fwd4<.,2,2,3,3,3> with 17 spilled VGPRs, conv_fwd3r<.,2,2,3,3,3> with 6 and
conv_fwd2<.,2,4,4> with 2134: a spill inside a persistent MFMA loop is a scratch round trip per tile."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _usage():
    from saragan_amd import build as b
    return b.resource_usage()


def test_no_kernel_spills_or_uses_scratch():
    usage = _usage()
    kernels = {k: v for f in usage.values() for k, v in f.items()}
    assert len(kernels) >= 100, 'resource table looks truncated'
    bad = {k: v for k, v in kernels.items() if v.get('vgpr_spill', 0) or v.get('scratch_bytes', 0)}
    assert not bad, f'kernels with spills / scratch: {json.dumps(bad, indent=1)}'
    for k, v in kernels.items():
        assert v['vgprs'] + v.get('agprs', 0) <= 512, k


def test_persistent_conv_kernels_fit_two_waves_per_simd():
    """The 512-thread ping-pong kernels run two waves per SIMD: at most 256 registers per lane."""
    usage = _usage()
    n = 0
    for k, v in usage['conv3d.hip'].items():
        if any(t in k for t in ('conv_fwd3r', 'conv_fwd3s', 'conv_fwd4', 'conv_fwd5')):
            n += 1
            assert v['vgprs'] + v.get('agprs', 0) <= 256, (k, v)
    assert n >= 20


def test_committed_resource_table_is_current():
    """profiles/r02_kernel_resources.json is the judged copy of the table: it must list every kernel of this build."""
    path = os.path.join(ROOT, 'profiles', 'r02_kernel_resources.json')
    committed = json.load(open(path))
    built = _usage()
    for f, ks in built.items():
        assert set(ks) == set(committed[f]), f'{f}: regenerate with python tools/dump_resources.py'
        for k, v in ks.items():
            assert committed[f][k]['vgpr_spill'] == v['vgpr_spill'] == 0
