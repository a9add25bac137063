"""GPU parity of each HIP kernel (through the C ABI via saragan_amd.functional) against the CPU oracle.
fp32 path: rtol 1e-4 / atol 1e-5 x output scale (f32 MFMA is an exact fmaf chain; only summation order
differs from the fp64 oracle).  bf16 path: inputs are bf16-rounded first, so the only error is f32
accumulation order plus one bf16 rounding of the output: rtol 1e-2 / atol 1e-2 x output scale."""
import numpy as np
import pytest
import torch

from oracle import pgan_oracle as O

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def tol(dtype):
    return (1e-4, 1e-5) if dtype == torch.float32 else (1e-2, 1e-2)


def rnd(shape, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g, dtype=torch.float64)
    return x.to(dtype).to(torch.float64)   # value representable in `dtype`


def cl(x, dtype):
    x = x.to(dtype).to(dev())
    return x.contiguous(memory_format=torch.channels_last_3d) if x.dim() == 5 else x.contiguous()


def close(got, ref, dtype, what=''):
    got = got.detach().double().cpu().numpy()
    ref = ref.detach().double().cpu().numpy()
    rt, at = tol(dtype)
    scale = max(1e-30, float(np.abs(ref).max()))
    np.testing.assert_allclose(got, ref, rtol=rt, atol=at * scale, err_msg=what)


CONV_CASES = [
    # n, cin, cout, (d,h,w), k
    (2, 16, 32, (4, 8, 8), (3, 3, 3)),
    (1, 32, 32, (4, 16, 32), (3, 3, 3)),
    (3, 8, 8, (1, 4, 4), (1, 3, 3)),
    (2, 24, 40, (3, 5, 7), (3, 3, 3)),       # ragged extents, channels not multiples of 16/32
    (2, 64, 96, (2, 8, 8), (3, 3, 3)),       # 3 N tiles, 4 K chunks
    (2, 32, 160, (2, 4, 4), (1, 3, 3)),      # 5 N tiles -> two blocks in y
    (4, 16, 16, (2, 4, 4), (1, 1, 1)),
    (2, 1, 16, (4, 8, 8), (1, 1, 1)),        # from_rgb shape
    (2, 16, 1, (4, 8, 8), (1, 1, 1)),        # to_rgb shape
    (2, 6, 10, (3, 5, 4), (3, 3, 3)),        # scalar load/store paths
    (1, 16, 16, (5, 20, 40), (3, 3, 3)),     # non power-of-two extents (start shape (1,5,16,16) family)
    (2, 32, 64, (16, 64, 64), (3, 3, 3)),    # >= 512 tiles: persistent weight-stationary kernel, 2 cout slices
    (2, 16, 32, (17, 66, 70), (3, 3, 3)),    # same kernel, ragged extents, single K chunk
    (2, 48, 96, (16, 64, 64), (3, 3, 3)),    # streamed-weight ping-pong kernel: 3 K chunks, 3 cout tiles over 2 blocks
    (4, 64, 32, (2, 64, 64), (1, 3, 3)),     # same, (1,3,3) taps, one cout tile
    (2, 32, 32, (6, 128, 256), (3, 3, 3)),   # 512 tile columns: sliding-halo wgrad (ring of D planes), ragged D tiles
]


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv3d_fwd_dgrad_wgrad(case, dtype):
    from saragan_amd import functional as F
    n, cin, cout, sp, k = case
    x = rnd((n, cin, *sp), 1, dtype)
    w = rnd((*k, cin, cout), 2, dtype)
    gy = rnd((n, cout, *sp), 3, dtype)
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef if dtype == torch.bfloat16 else w   # kernel rounds coef*w
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    yr = O.conv3d(xr, wr, 'leaky_relu', 0.2)
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], gy)

    xg = cl(x, dtype).requires_grad_(True)
    wg = w.float().to(dev()).requires_grad_(True)
    yg = F.conv3d(xg, wg, coef)
    close(yg, yr, dtype, 'fwd')
    gxg, gwg = torch.autograd.grad(yg, [xg, wg], cl(gy, dtype))
    gxr2 = gxr
    if dtype == torch.bfloat16:   # dgrad kernel rounds coef*w too: same wq
        pass
    close(gxg, gxr2, dtype, 'dgrad')
    rt, at = (1e-4, 1e-5) if dtype == torch.float32 else (2e-3, 2e-3)   # wgrad output is f32 in both paths
    ref = gwr.numpy()
    np.testing.assert_allclose(gwg.double().cpu().numpy(), ref, rtol=rt, atol=at * np.abs(ref).max(), err_msg='wgrad')


@pytest.mark.parametrize('dtype', DT)
def test_conv3d_fused_epilogue_and_upsample(dtype):
    from saragan_amd import functional as F
    n, cin, cout, sp, k = 2, 16, 32, (2, 4, 4), (3, 3, 3)
    x = rnd((n, cin, *sp), 4, dtype)
    w = rnd((*k, cin, cout), 5, dtype)
    b = rnd((cout,), 6, torch.float32) * 0.5
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = O.pixel_norm(O.act(O.apply_bias(O.conv3d(O.upscale3d(xr), wr, 'leaky_relu', 0.2), br), 'leaky_relu', 0.2))
    gy = rnd(tuple(yr.shape), 7, dtype)
    gr = torch.autograd.grad(yr, [xr, wr, br], gy)
    for fuse in (True, False):
        xg = cl(x, dtype).requires_grad_(True)
        wg = w.float().to(dev()).requires_grad_(True)
        bg = b.float().to(dev()).requires_grad_(True)
        if fuse:
            yg = F.conv3d(xg, wg, coef, bias=bg, act=True, slope=0.2, pixel_norm=True, upsample_in=True)
        else:
            yg = F.pixel_norm(F.bias_act(F.conv3d(F.upscale2x(xg), wg, coef), bg, True, 0.2))
        close(yg, yr, dtype, f'fwd fuse={fuse}')
        gg = torch.autograd.grad(yg, [xg, wg, bg], cl(gy, dtype))
        # the unfused bf16 chain rounds intermediates to bf16: widen the tolerance for it
        loose = (dtype == torch.bfloat16)
        for a, r, nm in zip(gg, gr, 'x w b'.split()):
            ref = r.numpy()
            rt, at = ((3e-2, 3e-2) if loose else (2e-4, 2e-5))
            np.testing.assert_allclose(a.double().cpu().numpy(), ref, rtol=rt, atol=at * np.abs(ref).max(),
                                       err_msg=f'grad {nm} fuse={fuse}')


@pytest.mark.parametrize('dtype', DT)
def test_dense(dtype):
    from saragan_amd import functional as F
    x = rnd((5, 48), 8, dtype)
    w = rnd((48, 20), 9, dtype)
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    xr, wr = x.clone().requires_grad_(True), wq.clone().requires_grad_(True)
    yr = O.dense(xr, wr, 'leaky_relu', 0.2)
    gy = rnd(tuple(yr.shape), 10, dtype)
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], gy)
    xg, wg = cl(x, dtype).requires_grad_(True), w.float().to(dev()).requires_grad_(True)
    yg = F.conv3d(xg, wg, coef)
    close(yg, yr, dtype)
    gxg, gwg = torch.autograd.grad(yg, [xg, wg], cl(gy, dtype))
    close(gxg, gxr, dtype)
    np.testing.assert_allclose(gwg.double().cpu().numpy(), gwr.numpy(), rtol=2e-3, atol=2e-3 * float(gwr.abs().max()))


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('c', [1, 8, 12, 32, 256])
def test_elementwise_ops(dtype, c):
    from saragan_amd import functional as F
    x = rnd((3, c, 2, 4, 6), 11, dtype)
    b = rnd((c,), 12, torch.float32)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xg, bg = cl(x, dtype).requires_grad_(True), b.float().to(dev()).requires_grad_(True)
    gy = rnd(tuple(x.shape), 13, dtype)
    # bias + leaky relu
    yr = O.act(O.apply_bias(xr, br), 'leaky_relu', 0.2)
    yg = F.bias_act(xg, bg, True, 0.2)
    close(yg, yr, dtype, 'bias_act')
    gr = torch.autograd.grad(yr, [xr, br], gy)
    gg = torch.autograd.grad(yg, [xg, bg], cl(gy, dtype))
    close(gg[0], gr[0], dtype, 'bias_act dx')
    np.testing.assert_allclose(gg[1].double().cpu().numpy(), gr[1].numpy(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=1e-2 * float(gr[1].abs().max()) if dtype == torch.bfloat16 else 1e-5)
    # pixel norm
    yr = O.pixel_norm(xr)
    yg = F.pixel_norm(xg)
    close(yg, yr, dtype, 'pixel_norm')
    (gr0,) = torch.autograd.grad(yr, xr, gy)
    (gg0,) = torch.autograd.grad(yg, xg, cl(gy, dtype))
    rt, at = (1e-4, 1e-5) if dtype == torch.float32 else (3e-2, 3e-2)   # bwd uses the bf16-rounded y
    if c > 1:   # c == 1: y = sign(x), the gradient is pure cancellation (~1e-8 |dy|) and not meaningful
        np.testing.assert_allclose(gg0.double().cpu().numpy(), gr0.numpy(), rtol=rt, atol=at * float(gr0.abs().max()))
    # up / down / lerp
    close(F.upscale2x(xg), O.upscale3d(xr), dtype, 'up')
    close(F.downscale2x(xg), O.downscale3d(xr), dtype, 'down')
    (gu,) = torch.autograd.grad(F.upscale2x(xg), xg, cl(O.upscale3d(gy), dtype))
    close(gu, 8 * gy, dtype, 'up bwd')
    (gd,) = torch.autograd.grad(F.downscale2x(xg), xg, cl(O.downscale3d(gy), dtype))
    close(gd, O.upscale3d(O.downscale3d(gy)) / 8, dtype, 'down bwd')
    close(F.lerp(xg, cl(gy, dtype), 0.3, 0.7), 0.3 * x + 0.7 * gy, dtype, 'lerp')


@pytest.mark.parametrize('dtype', DT)
def test_sumsq_keep_w_and_mbstd(dtype):
    from saragan_amd import functional as F
    g = rnd((4, 1, 4, 8, 16), 14, dtype)
    ref = (g * g).sum(dim=(1, 2, 3))
    got = F.sumsq_keep_w(cl(g, dtype))
    np.testing.assert_allclose(got.double().cpu().numpy(), ref.numpy(), rtol=1e-4)
    x = rnd((8, 6, 1, 4, 4), 15, dtype)
    close(F.minibatch_stddev(cl(x, dtype)), O.minibatch_stddev_layer(x), dtype, 'mbstd')


def test_add_noise_statistics():
    from saragan_amd import functional as F
    x = torch.zeros(1, 1, 16, 64, 64, device=dev())
    a = F.add_noise(x, 2.0, seed=7)
    b = F.add_noise(x, 2.0, seed=7)
    c = F.add_noise(x, 2.0, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.mean())) < 0.05 and abs(float(a.std()) - 2.0) < 0.05
    k = float(((a / 2.0) ** 4).mean())
    assert abs(k - 3.0) < 0.2   # gaussian kurtosis


def test_adam_ema_matches_tf_rule():
    from saragan_amd import functional as F
    n = 1003
    g = torch.Generator().manual_seed(16)
    p0 = torch.randn(n, generator=g, dtype=torch.float64)
    params = {'w': p0.clone()}
    shadow = {'w': p0.clone()}
    opt = O.TFAdam(0.0, 0.9)
    p = p0.float().to(dev()); m = torch.zeros_like(p); v = torch.zeros_like(p); ema = p.clone()
    for step in range(1, 4):
        gr = torch.randn(n, generator=g, dtype=torch.float64)
        opt.apply(params, {'w': gr}, 1e-3)
        O.ema_update(shadow, params, 0.99)
        F.adam_ema_(p, gr.float().to(dev()), m, v, ema, 1e-3, 0.0, 0.9, step)
    np.testing.assert_allclose(p.double().cpu().numpy(), params['w'].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ema.double().cpu().numpy(), shadow['w'].numpy(), rtol=1e-5, atol=1e-6)
