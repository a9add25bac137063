"""The GEMM-tiled convolution of the low-resolution levels (csrc/gemm.hip: 256 voxels of the folded batch x 128 output
channels per block, K split over blocks where the level is too small to fill the chip) against the spatial kernels
(SG_NO_GEMM=1) and the fp64 oracle: the shapes of pgan 's' at 1x4x4, 2x8x8 and 4x16x16 (pgan/generator.py:26-45,
pgan/discriminator.py:48-68) at batch 32 / 64, forward with bias + LeakyReLU + sign words, the data gradient with a
LeakyReLU mask, the fused nearest-x2 gather of the generator's conv_1.  The kernel name is asserted."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pgan_oracle as O

pytestmark = pytest.mark.gpu

CASES = [
    # n, cin, cout, (d, h, w), kernel, upsample_in, expects the K-split variant
    (32, 512, 512, (2, 8, 8), (1, 3, 3), False, True),
    (64, 512, 512, (2, 8, 8), (1, 3, 3), False, True),
    (32, 512, 512, (1, 4, 4), (1, 3, 3), False, True),
    (64, 512, 512, (1, 4, 4), (1, 3, 3), False, True),
    (256, 128, 256, (2, 8, 8), (1, 3, 3), False, False),      # enough tiles without a K split
    (32, 512, 512, (2, 8, 8), (1, 3, 3), True, True),
]


def _kernels(lib, _lib):
    ents = (_lib.ProfEntry * 16)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 16, C.byref(cnt))
    return sorted({ents[i].kernel.decode() for i in range(cnt.value)})


@pytest.mark.parametrize('case', CASES, ids=[f'n{c[0]}_{c[1]}to{c[2]}at{"x".join(map(str, c[3]))}{"_ups" if c[5] else ""}' for c in CASES])
def test_gemm_conv_matches_spatial_kernels_and_oracle(case, sg_env):
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp, k, ups, split = case
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(cin + cout + sp[1])
    in_sp = tuple(s // 2 for s in sp) if ups else sp
    x = torch.randn((n, cin, *in_sp), generator=g).bfloat16()
    w = torch.randn((*k, cin, cout), generator=g)
    b = torch.randn(cout, generator=g) * 0.1
    gy = torch.randn((n, cout, *sp), generator=g).bfloat16()
    coef = float(O.runtime_coef(w.shape, 'leaky_relu', 0.2))
    xd = x.to(dev).contiguous(memory_format=torch.channels_last_3d)
    gd = gy.to(dev).contiguous(memory_format=torch.channels_last_3d)
    wd, bd = w.to(dev), b.to(dev)

    def run():
        lib.sg_prof_enable(1)
        y, _, signs = F.raw_conv(xd, wd, coef, False, ups, bias=bd, act=True, slope=0.2, want_signs=True)
        out = [y, signs]
        if not ups:       # data gradient of this layer with the LeakyReLU mask of its input in the epilogue
            gx, _, _ = F.raw_conv(gd, wd, coef, True, mask_bits=F.sign_words(xd), mask_slope=0.2)
            out.append(gx)
        torch.cuda.synchronize()
        kern = _kernels(lib, _lib)
        lib.sg_prof_enable(0)
        F.clear_pack_cache()
        return out, kern

    got, kern = run()
    # (the data gradient has the channel counts swapped and may split differently: every launch is a GEMM kernel, and the
    # forward's variant is the expected one)
    assert set(kern) <= {'conv_gemm', 'conv_gemm (K split)'} and ('conv_gemm (K split)' if split else 'conv_gemm') in kern, kern
    sg_env(SG_NO_GEMM=1)
    ref, kern_ref = run()
    assert not any('gemm' in k_ for k_ in kern_ref), kern_ref
    for name, a_, r_ in zip(('y', 'sign words', 'masked data gradient'), got, ref):
        if a_.dtype == torch.int32:
            assert float((a_ != r_).float().mean()) <= 5e-3, name
            continue
        err = float((a_.double() - r_.double()).abs().max() / r_.double().abs().max())
        assert err <= 1e-2, (name, err)
    # against the fp64 oracle (coef * w rounded to bf16, as the packed image holds it), samples 0 and n-1
    wq = ((w * coef).bfloat16().double() / coef)
    for smp in (0, n - 1):
        xs = x[smp:smp + 1].double()
        if ups:
            xs = O.upscale3d(xs)
        yr = O.act(O.apply_bias(O.conv3d(xs, wq, 'leaky_relu', 0.2), b.double()), 'leaky_relu', 0.2)
        err = float((got[0][smp:smp + 1].double().cpu() - yr).abs().max() / yr.abs().max())
        assert err <= 1e-2, ('oracle', smp, err)
