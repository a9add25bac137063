"""The GEMM-tiled convolution of the low-resolution levels (csrc/gemm.hip: 256 voxels of the folded batch x 128 output
channels per block, K split over blocks where the level is too small to fill the chip) against the spatial kernels
(SG_NO_GEMM=1) and the fp64 oracle: the shapes of pgan 's' at 1x4x4, 2x8x8 and 4x16x16 (pgan/generator.py:26-45,
pgan/discriminator.py:48-68) at batch 32 / 64 and at the batches of 1-5 the reference's own batch rule gives (tiles with
empty sample slots; the 3x3x3 layers of the 4x16x16 level), forward with bias + LeakyReLU + sign words, the data gradient with a
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
    # small batches (round 5): sample slots of a tile beyond the batch; the 27-tap layers of the 4x16x16 level, one plane per tile
    (2, 512, 512, (1, 4, 4), (1, 3, 3), False, True),         # 2 of a tile's 16 sample slots
    (5, 256, 256, (1, 4, 4), (1, 3, 3), False, True),
    (19, 128, 128, (1, 4, 4), (1, 3, 3), False, True),        # a whole tile and 3 slots of the next
    (3, 512, 512, (2, 8, 8), (1, 3, 3), False, True),         # sample pairs: the last tile half empty
    (1, 256, 256, (2, 8, 8), (1, 3, 3), True, True),
    (2, 128, 128, (4, 16, 16), (3, 3, 3), False, True),
    (2, 128, 512, (4, 16, 16), (3, 3, 3), False, None),       # (the K split follows the partial-tile cost: either variant)
    (2, 512, 128, (4, 16, 16), (3, 3, 3), False, True),
    (1, 512, 128, (4, 16, 16), (3, 3, 3), True, True),        # the generator's conv_1 of that level: nearest-x2 gather fused
    (4, 128, 128, (4, 16, 16), (3, 3, 3), False, None),
    (3, 256, 128, (4, 16, 16), (3, 3, 3), False, None),
    (2, 256, 64, (4, 16, 16), (3, 3, 3), False, None),        # 64 output channels: half of the block's N waves idle
    (2, 256, 64, (4, 16, 16), (3, 3, 3), True, None),
    (4, 128, 192, (2, 8, 8), (1, 3, 3), False, None),         # 192 = 128 + 64: a whole block and a half one
]


def _kernels(lib, _lib):
    ents = (_lib.ProfEntry * 16)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 16, C.byref(cnt))
    return sorted({ents[i].kernel.decode() for i in range(cnt.value)})


@pytest.mark.parametrize('case', CASES, ids=[f'n{c[0]}_{c[1]}to{c[2]}at{"x".join(map(str, c[3]))}{"_ups" if c[5] else ""}' for c in CASES])
def test_gemm_conv_matches_spatial_kernels_and_oracle(case, sg_env, monkeypatch):
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp, k, ups, split = case
    lib = _lib.load()
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)      # (the sub-pixel form would take a 27-tap up-sampled layer first)
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
    assert set(kern) <= {'conv_gemm', 'conv_gemm (K split)'} and kern, kern
    if split is not None:
        assert ('conv_gemm (K split)' if split else 'conv_gemm') in kern, kern
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


WGRAD_CASES = [
    # n, cin, cout, (d, h, w)
    (32, 512, 512, (2, 8, 8)),        # the 512 -> 512 layers of the 2x8x8 level: two planes per tile, 16 tiles per block
    (8, 512, 512, (1, 4, 4)),         # 1x4x4: eight samples per tile, ONE tile per block
    (3, 64, 96, (2, 8, 8)),           # fewer tiles than ring slots; cin != cout
    (24, 128, 64, (1, 4, 4)),
]


@pytest.mark.parametrize('case', WGRAD_CASES, ids=[f'n{c[0]}_{c[1]}to{c[2]}at{"x".join(map(str, c[3]))}' for c in WGRAD_CASES])
def test_plane_tiled_weight_gradient_matches_the_generic_kernel_and_torch(case, sg_env):
    """conv_wgrad_planes (csrc/wgrad.hip: whole (n, d) planes as tiles through a three-slot LDS ring, the four waves split
    a tile's K steps) against the generic kernel (SG_NO_GEMM=1) and torch's fp64 convolution backward; weight and bias
    gradient; the kernel name is asserted; reproducible mode gives the same bits twice."""
    import saragan_amd
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp = case
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(cin * 3 + cout + sp[1])
    x = torch.randn((n, cin, *sp), generator=g).bfloat16()
    gy = torch.randn((n, cout, *sp), generator=g).bfloat16()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last_3d)
    gd = gy.to(dev).contiguous(memory_format=torch.channels_last_3d)
    coef = 0.37

    def run():
        lib.sg_prof_enable(1)
        dw, db = F.raw_wgrad(xd, gd, (1, 3, 3), coef, False, True)
        torch.cuda.synchronize()
        kern = _kernels(lib, _lib)
        lib.sg_prof_enable(0)
        return dw, db, kern

    dw, db, kern = run()
    assert any(k.startswith('conv_wgrad_planes') for k in kern), kern
    # torch, fp64: dw[kh, kw, ci, co] = sum_v x[v + tap, ci] gy[v, co] per D plane
    x2 = x.double().permute(0, 2, 1, 3, 4).reshape(-1, cin, sp[1], sp[2])
    g2 = gy.double().permute(0, 2, 1, 3, 4).reshape(-1, cout, sp[1], sp[2])
    wz = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    (gw,) = torch.autograd.grad(torch.nn.functional.conv2d(x2, wz, padding=1), wz, g2)
    ref = gw.permute(2, 3, 1, 0).unsqueeze(0) * coef
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-5, err                       # f32 accumulation of bf16 products over <= 4096 voxels
    refb = gy.double().sum(dim=(0, 2, 3, 4))
    assert float((db.double().cpu() - refb).abs().max() / refb.abs().max()) <= 2e-5
    sg_env(SG_NO_GEMM=1)
    dw2, db2, kern2 = run()
    assert not any('planes' in k for k in kern2), kern2
    assert float((dw - dw2).abs().max() / dw2.abs().max()) <= 2e-5
    sg_env(SG_NO_GEMM=0)
    saragan_amd.set_deterministic(True)
    try:
        a1, b1, k1 = run()
        a2, b2, _ = run()
        assert any(k.startswith('conv_wgrad_planes') for k in k1), k1
        assert torch.equal(a1, a2) and torch.equal(b1, b2)
        assert float((a1 - dw).abs().max() / dw.abs().max()) <= 2e-5
    finally:
        saragan_amd.set_deterministic(False)
