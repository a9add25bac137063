"""GPU parity of each HIP kernel (through the C ABI via saragan_amd.functional) against the CPU oracle.
fp32 path: rtol 1e-4 / atol 1e-5 x output scale (f32 MFMA is an exact fmaf chain; only summation order
differs from the fp64 oracle).  bf16 path: inputs are bf16-rounded first, so the only error is f32
accumulation order plus one bf16 rounding of the output: rtol 1e-2 / atol 1e-2 x output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

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
    (2, 32, 32, (6, 128, 256), (3, 3, 3)),   # 512 tile columns: sliding-halo wgrad and forward (ring of D planes)
    (2, 16, 32, (5, 126, 256), (3, 3, 3)),   # sliding-halo forward, one 32-byte chunk, ragged D and H
    (2, 32, 64, (4, 64, 256), (3, 3, 3)),    # sliding-halo forward, two cout slices
    (32, 128, 128, (2, 8, 8), (1, 3, 3)),    # ping-pong wgrad on 8-wide tiles spanning two samples (TN = 2)
    (16, 64, 128, (4, 16, 16), (3, 3, 3)),   # ping-pong wgrad on 16-wide tiles
    (2, 32, 32, (1, 128, 256), (1, 3, 3)),   # 2-D image shapes (SURFGAN_2D, D = 1): persistent kernels, (1,3,3) taps
    (2, 3, 16, (1, 16, 16), (1, 1, 1)),      # 2-D from_rgb on RGB images
    (2, 16, 3, (1, 16, 16), (1, 1, 1)),      # 2-D to_rgb
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


def _ref_sign_words(t):
    """CPU restatement of the sign-word layout (include/saragan_hip.h): int32 [n,d,h,w,ceil(c/32)]."""
    t = t.permute(0, 2, 3, 4, 1).contiguous()
    c = t.shape[-1]
    nw = (c + 31) // 32
    neg = torch.zeros((*t.shape[:-1], nw * 32), dtype=torch.int64)
    neg[..., :c] = (t < 0).to(torch.int64)
    words = (neg.reshape(*t.shape[:-1], nw, 32) << torch.arange(32)).sum(-1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)


MASK_CASES = [
    (2, 16, 32, (4, 8, 8), (3, 3, 3)),       # generic kernels
    (2, 24, 40, (3, 5, 7), (3, 3, 3)),       # padded-row fallback kernel, ragged
    (2, 16, 1, (4, 8, 8), (1, 1, 1)),        # point-wise weight gradient with the ones channel
    (2, 1, 16, (4, 8, 8), (1, 1, 1)),
    (2, 32, 32, (16, 64, 64), (3, 3, 3)),    # persistent weight-stationary kernel; sliding-halo wgrad with the ones tap
    (2, 64, 32, (8, 64, 64), (3, 3, 3)),     # streamed ping-pong kernel, one cout tile; 2 cin tiles in wgrad
    (2, 64, 64, (8, 32, 64), (3, 3, 3)),     # streamed ping-pong kernel, two cout tiles
    (2, 32, 32, (5, 128, 256), (3, 3, 3)),   # sliding-halo forward kernel (mask words, sign_out), ragged D
]


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('case', MASK_CASES)
def test_conv_mask_epilogue_and_wgrad_bias(case, dtype):
    """sg_conv_epilogue.mask_bits / sign_out (LeakyReLU backward fused into the data-gradient conv through sign
    words) and sg_conv3d_wgrad_bias (bias gradient from the weight-gradient kernel)."""
    from saragan_amd import functional as F
    n, cin, cout, sp, k = case
    x = rnd((n, cin, *sp), 11, dtype)
    w = rnd((*k, cin, cout), 12, dtype)
    m = rnd((n, cout, *sp), 13, dtype)
    m[:, :, 0, 0, :3] = 0.0                      # a >= 0 counts as the positive side (where(a >= 0, ...))
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef if dtype == torch.bfloat16 else w
    ref = O.conv3d(x, wq, 'leaky_relu', 0.2) * torch.where(m >= 0, 1.0, 0.2)
    mbits = F.sign_words(cl(m, dtype))
    ref_words = _ref_sign_words(m)
    assert torch.equal(mbits.cpu(), ref_words), 'sg_sign_words'
    got, _, _ = F.raw_conv(cl(x, dtype), w.float().to(dev()), coef, False, False, mask_bits=mbits, mask_slope=0.2)
    close(got, ref, dtype, 'masked fwd')
    # sign words written by the epilogue itself (bias + LeakyReLU layer): must equal the signs of the stored output
    b = rnd((cout,), 16, torch.float32) * 0.5
    yb, _, sg = F.raw_conv(cl(x, dtype), w.float().to(dev()), coef, False, False, bias=b.float().to(dev()), act=True,
                           slope=0.2, want_signs=True)
    pre = O.apply_bias(O.conv3d(x, wq, 'leaky_relu', 0.2), b.double())
    sure = (pre.abs() > 1e-3 * pre.abs().max())                 # away from zero the sign is not a rounding matter
    diff = (_ref_sign_words(yb.double().cpu()) ^ sg.cpu())
    assert int(diff.ne(0).sum()) == 0, 'sign_out differs from the signs of the stored activation'
    assert torch.equal((yb.double().cpu() < 0) & sure, (pre < 0) & sure)
    # flipped (data-gradient) weights with the mask, as the backward pass uses it
    gy = rnd((n, cout, *sp), 14, dtype)
    mx = rnd((n, cin, *sp), 15, dtype)
    xr = x.clone().requires_grad_(True)
    (gxr,) = torch.autograd.grad(O.conv3d(xr, wq, 'leaky_relu', 0.2), [xr], gy)
    got, _, _ = F.raw_conv(cl(gy, dtype), w.float().to(dev()), coef, True, False, mask_bits=F.sign_words(cl(mx, dtype)),
                           mask_slope=0.2)
    close(got, gxr * torch.where(mx >= 0, 1.0, 0.2), dtype, 'masked dgrad')
    # weight + bias gradient in one call
    dw, db = F.raw_wgrad(cl(x, dtype), cl(gy, dtype), k, coef, False, want_db=True)
    wr = wq.clone().requires_grad_(True)
    (gwr,) = torch.autograd.grad(O.conv3d(x, wr, 'leaky_relu', 0.2), [wr], gy)
    rt, at = (1e-4, 1e-5) if dtype == torch.float32 else (2e-3, 2e-3)
    np.testing.assert_allclose(dw.double().cpu().numpy(), gwr.numpy(), rtol=rt, atol=at * gwr.abs().max().item())
    dbr = gy.sum((0, 2, 3, 4)).numpy()
    np.testing.assert_allclose(db.double().cpu().numpy(), dbr, rtol=rt, atol=at * max(1.0, np.abs(dbr).max()))


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('shape', [(2, 16, 3, 4, 5), (2, 64, 2, 3, 16)])   # flat kernel / row-wise kernel
def test_upscale2x_masked(shape, dtype):
    from saragan_amd import functional as F
    n_, c_, d_, h_, w_ = shape
    x = rnd(shape, 21, dtype)
    m = rnd((n_, c_, 2 * d_, 2 * h_, 2 * w_), 22, dtype)
    ref = 0.125 * O.upscale3d(x) * torch.where(m >= 0, 1.0, 0.2)
    got = F._Up.apply(cl(x, dtype), 0.125, F.sign_words(cl(m, dtype)), 0.2)
    close(got, ref, dtype)
    # the stand-alone LeakyReLU backward with sign words instead of the activation
    g = rnd(tuple(m.shape), 23, dtype)
    dx, db = F.raw_bias_act_bwd(cl(g, dtype), F.sign_words(cl(m, dtype)), 0.2, True, True)
    refdx = g * torch.where(m >= 0, 1.0, 0.2)
    close(dx, refdx, dtype)
    np.testing.assert_allclose(db.double().cpu().numpy(), refdx.sum((0, 2, 3, 4)).numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('big', [False, True])
def test_fused_mask_chain_first_and_second_order(big, dtype):
    """Discriminator-block shaped chain conv-bias-lrelu -> conv-bias-lrelu -> downscale built through the ops layer,
    where the LeakyReLU backward of each layer runs inside its consumer's backward kernel: first-order gradients and
    the gradient-penalty style second-order gradients against the oracle."""
    from saragan_amd import varstore
    from saragan_amd.networks import ops
    n, c0, c1, c2, sp = (2, 32, 32, 64, (8, 64, 64)) if big else (2, 8, 16, 16, (2, 4, 8))
    x = rnd((n, c0, *sp), 31, dtype) 
    w1 = rnd((3, 3, 3, c0, c1), 32, dtype)
    w2 = rnd((3, 3, 3, c1, c2), 33, dtype)
    b1 = rnd((c1,), 34, torch.float32) * 0.3
    b2 = rnd((c2,), 35, torch.float32) * 0.3

    def stored(h):   # the kernels store activations in `dtype`: same values (hence the same LeakyReLU masks) here
        return h + (h.detach().to(dtype).double() - h.detach())

    def ref_chain(xr, w1r, w2r, b1r, b2r):
        h = stored(O.act(O.apply_bias(O.conv3d(xr, w1r, 'leaky_relu', 0.2), b1r), 'leaky_relu', 0.2))
        h = stored(O.act(O.apply_bias(O.conv3d(h, w2r, 'leaky_relu', 0.2), b2r), 'leaky_relu', 0.2))
        return O.downscale3d(h)

    c1f = O.runtime_coef(w1.shape, 'leaky_relu', 0.2)
    c2f = O.runtime_coef(w2.shape, 'leaky_relu', 0.2)
    q = (lambda w, c: (w * c).to(dtype).double() / c) if dtype == torch.bfloat16 else (lambda w, c: w)
    rv = [t.clone().requires_grad_(True) for t in (x, q(w1, c1f), q(w2, c2f), b1.double(), b2.double())]
    yr = ref_chain(*rv)
    gy = rnd(tuple(yr.shape), 36, dtype)
    g1r = torch.autograd.grad(yr, rv, gy, create_graph=True)
    pen_r = (g1r[0] ** 2).sum()
    g2r = torch.autograd.grad(pen_r, rv[1:], allow_unused=True)

    store = varstore.VariableStore(dev())
    old_dt = varstore.compute_dtype()
    varstore.set_compute_dtype(dtype)

    def layer(name, h, fmaps):
        with varstore.variable_scope(name):
            return ops.act(ops.apply_bias(ops.conv3d(h, fmaps, (3, 3, 3), 'leaky_relu', param=0.2)), 'leaky_relu', 0.2)

    try:
        with varstore.use_store(store):
            xg = cl(x, dtype).requires_grad_(True)
            layer('l2', layer('l1', xg, c1), c2)         # creates the variables
            store.load_state_dict({'l1/weight': w1, 'l2/weight': w2, 'l1/bias': b1, 'l2/bias': b2}, strict=True)
            yg = ops.materialize(ops.downscale3d(layer('l2', layer('l1', xg, c1), c2)))
        pv = store.vars
        gv = [xg, pv['l1/weight'], pv['l2/weight'], pv['l1/bias'], pv['l2/bias']]
        close(yg, yr, dtype, 'chain fwd')
        g1 = torch.autograd.grad(yg, gv, cl(gy, dtype), create_graph=True)
        pen = (g1[0].float() ** 2).sum()
        g2 = torch.autograd.grad(pen, gv[1:], allow_unused=True)
    finally:
        varstore.set_compute_dtype(old_dt)
    rt, at = (2e-4, 2e-5) if dtype == torch.float32 else (3e-2, 3e-2)
    for a, r, nm in zip(g1, g1r, 'x w1 w2 b1 b2'.split()):
        ref = r.detach().numpy()
        np.testing.assert_allclose(a.detach().double().cpu().numpy(), ref, rtol=rt, atol=at * np.abs(ref).max(),
                                   err_msg=f'first-order {nm}')
    for a, r, nm in zip(g2, g2r, 'w1 w2 b1 b2'.split()):
        if r is None or a is None:   # the penalty is piecewise constant in the biases: "unused" or exactly zero
            for t in (a, r):
                assert t is None or float(t.abs().max()) == 0.0, nm
            continue
        ref = r.detach().numpy()
        np.testing.assert_allclose(a.detach().double().cpu().numpy(), ref, rtol=rt, atol=at * np.abs(ref).max(),
                                   err_msg=f'second-order {nm}')


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cout', [32, 64])
def test_upconv_fused_gather_pingpong_kernel(cout, dtype, sg_env):
    """conv3d(upscale3d(x)) with Cin = 64 through the streamed ping-pong kernel's buffer-addressed x2 gather
    (boundary and interior tiles) against the oracle."""
    from saragan_amd import functional as F
    sg_env(SG_FWD4_GX=8)       # reach the ping-pong kernel with a small tensor (>= 16 tiles)
    n, cin, sp = 2, 64, (4, 8, 32)
    x = rnd((n, cin, *sp), 51, dtype)
    w = rnd((3, 3, 3, cin, cout), 52, dtype)
    b = rnd((cout,), 53, torch.float32) * 0.5
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    ref = O.act(O.apply_bias(O.conv3d(O.upscale3d(x), wq, 'leaky_relu', 0.2), b.double()), 'leaky_relu', 0.2)
    F.clear_pack_cache()
    y, _, _ = F.raw_conv(cl(x, dtype), w.float().to(dev()), coef, False, True, bias=b.float().to(dev()), act=True,
                         slope=0.2)
    close(y, ref, dtype, 'fused x2 gather')


SUBPIX_DGRAD_CASES = [
    # n, ci (channels of x), co (channels of gy), low-resolution (d, h, w)
    (2, 64, 32, (4, 8, 32)),          # the 64 -> 32 layer: tile 2 x 4 x 32, two chunks
    (1, 128, 64, (2, 8, 64)),         # two 64-channel parts of gx (blockIdx.y), four chunks, two W tiles
    (3, 128, 128, (2, 8, 16)),        # 16-wide rows: tile 2 x 8 x 16; more tiles than one block takes in a trip
    (1, 64, 48, (6, 4, 32)),          # co not a multiple of 32; three D tiles
]


@pytest.mark.parametrize('case', SUBPIX_DGRAD_CASES, ids=[f'{c[1]}from{c[2]}at{"x".join(map(str, c[3]))}' for c in SUBPIX_DGRAD_CASES])
def test_upconv_subpixel_data_gradient(case, monkeypatch):
    """Gradient of conv3d(upscale3d(x)) for x in sub-pixel form (sg_upconv3d_subpixel_dgrad: a stride-2, 4 x 4 x 4-tap
    convolution of the fine gradient with the forward's summed weights transposed) against (a) the same sum in fp64 with
    the weights the packed image holds, (b) the fp64 oracle's autograd through ops.py:276-289 + :147-150 and (c) the library's
    27-tap pooled path; the kernel name is asserted."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, ci, co, sp = case
    dtype = torch.bfloat16
    fine = tuple(2 * v for v in sp)
    w = rnd((3, 3, 3, ci, co), 71, dtype)
    gy = rnd((n, co, *fine), 72, dtype)
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wg, gyg = w.float().to(dev()), cl(gy, dtype)
    F.clear_pack_cache()
    lib = _lib.load()
    lib.sg_prof_enable(1)
    with torch.no_grad():
        gx = F._upconv_dgrad_subpixel(gyg, wg, coef)
    torch.cuda.synchronize()
    ents = (_lib.ProfEntry * 8)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 8, C.byref(cnt))
    lib.sg_prof_enable(0)
    assert gx is not None, 'the sub-pixel data gradient was not taken'
    assert [ents[i].kernel.decode() for i in range(cnt.value)] == ['upconv_subpixel_dgrad']
    assert tuple(gx.shape) == (n, ci, *sp)
    # (a) the kernel's own arithmetic in fp64: W4[j] = bf16(coef * sum of the taps folded into j), j = -1: {2}, 0: {1, 2}, 1: {0, 1}, 2: {0}
    sets = [(2,), (1, 2), (0, 1), (0,)]
    wd = w.double()
    w4 = torch.zeros((4, 4, 4, ci, co), dtype=torch.float64)
    for a_ in range(4):
        for b_ in range(4):
            for c_ in range(4):
                acc = 0
                for kd in sets[a_]:
                    for kh in sets[b_]:
                        for kw in sets[c_]:
                            acc = acc + wd[kd, kh, kw]
                w4[a_, b_, c_] = (acc * coef).to(dtype).double()
    ref_a = torch.nn.functional.conv3d(gy.double(), w4.permute(3, 4, 0, 1, 2), stride=2, padding=1)
    close(gx, ref_a, dtype, 'sub-pixel dgrad vs its own sum in fp64')
    # (b) autograd of the reference formulation, fp64, per-tap rounded weights (what the forward multiplies with)
    xr = torch.zeros((n, ci, *sp), dtype=torch.float64, requires_grad=True)
    wq = (w.double() * coef).to(dtype).double() / coef
    yr = O.conv3d(O.upscale3d(xr), wq, 'leaky_relu', 0.2)
    (ref_b,) = torch.autograd.grad(yr, xr, gy.double())
    err = float(torch.linalg.vector_norm(gx.double().cpu() - ref_b) / torch.linalg.vector_norm(ref_b))
    assert err <= 1e-2, err          # bf16 rounding of the summed weights (2^-9 relative per weight) and of the result
    # (c) the 27-tap pooled path of the library
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)
    with torch.no_grad():
        gx2 = F._upconv_dgrad(gyg, wg, coef, True)
    err2 = float(torch.linalg.vector_norm(gx.double() - gx2.double()) / torch.linalg.vector_norm(gx2.double()))
    assert err2 <= 1e-2, err2


def test_subpixel_kernels_are_mutually_adjoint_at_the_benchmarked_size():
    """Size-independent properties at the size the bench runs (batch 32, 64 -> 32 channels, 16x64x64 -> 32x128x128), where
    the oracle cannot follow: the sub-pixel forward F (no bias, no activation), its data gradient D and its weight
    gradient W are three views of one bilinear form, so <F(x; w), g> = <x, D(g; w)> = <w, W(x, g)>.  F and D multiply with
    the same rounded summed weights (their identity holds to the bf16 rounding of the outputs); W does not see the weights
    (its identity holds to the rounding of the summed weights, 2^-9 relative each)."""
    from saragan_amd import functional as F
    n, cin, cout, sp = 32, 64, 32, (16, 64, 64)
    g_ = torch.Generator(device='cuda').manual_seed(91)
    x = torch.randn((n, cin, *sp), generator=g_, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((3, 3, 3, cin, cout), generator=g_, device='cuda')
    coef = 0.024
    with torch.no_grad():
        res = F._raw_upconv_subpixel(x, w, coef, None, False, 0.0, False, 1e-8, False, False)
        assert res is not None, 'the sub-pixel forward was not taken'
        y = res[0]
        # g = F(x) + noise: the form is then dominated by |F(x)|^2, far above what uncorrelated outputs would produce
        gy = (y.float() + 0.25 * torch.randn(y.shape, generator=g_, device='cuda').contiguous(memory_format=torch.channels_last_3d)).bfloat16()
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = F._upconv_dgrad_subpixel(gy, w, coef)
        assert gx is not None, 'the sub-pixel data gradient was not taken'
        dw, _ = F.raw_wgrad(x, gy, (3, 3, 3), coef, True, False)
        a = float((y.double() * gy.double()).sum())
        b = float((x.double() * gx.double()).sum())
        c = float((w.double() * dw.double()).sum())          # dw = coef * sum x (x) gy, F multiplies with coef * w: <w, dw> = <F, g>
        yy = float((y.double() * y.double()).sum())
    assert np.isfinite([a, b, c]).all() and a > 0.9 * yy, (a, yy)
    assert abs(a - b) <= 1e-3 * a, (a, b)
    assert abs(a - c) <= 2e-3 * a, (a, c)


SUBPIX_CASES = [
    # n, cin, cout, low-resolution (d, h, w), pixel_norm
    (2, 64, 32, (4, 8, 32), True),        # the 64 -> 32 layer's tile (2 x 4 x 32), pixel-norm in the epilogue
    (1, 32, 64, (2, 8, 32), False),       # two output-channel tiles
    (2, 128, 64, (4, 16, 16), False),     # 16-wide rows: tile 2 x 8 x 16
    (1, 128, 64, (2, 8, 32), True),       # pixel-norm over 64 channels: the two-N-tile kernel
    (1, 48, 32, (8, 8, 8), True),         # 8-wide rows: tile 4 x 8 x 8; cin not a multiple of 32
]


@pytest.mark.parametrize('case', SUBPIX_CASES, ids=[f'{c[1]}to{c[2]}at{"x".join(map(str, c[3]))}' for c in SUBPIX_CASES])
def test_upconv_subpixel_matches_oracle(case, monkeypatch):
    """conv3d(upscale3d(x)) in sub-pixel form -- ONE launch over the eight parity classes (csrc/subpix.hip) with bias +
    LeakyReLU (+ pixel-norm) + sign words + scale in the scatter epilogue -- against the oracle's 27-tap formulation of
    ops.py:276-289 + :147-150, and against the library's own fused-gather path; the kernel name is asserted."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp, pn = case
    dtype = torch.bfloat16
    x = rnd((n, cin, *sp), 41, dtype)
    w = rnd((3, 3, 3, cin, cout), 42, dtype)
    b = rnd((cout,), 43, torch.float32) * 0.5
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    ref = O.act(O.apply_bias(O.conv3d(O.upscale3d(x), w, 'leaky_relu', 0.2), b.double()), 'leaky_relu', 0.2)
    if pn:
        ref = O.pixel_norm(ref)
    xg, wg, bg = cl(x, dtype), w.float().to(dev()), b.float().to(dev())
    F.clear_pack_cache()
    lib = _lib.load()
    lib.sg_prof_enable(1)
    res = F._raw_upconv_subpixel(xg, wg, coef, bg, True, 0.2, pn, 1e-8, True, True)
    torch.cuda.synchronize()
    ents = (_lib.ProfEntry * 8)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 8, C.byref(cnt))
    lib.sg_prof_enable(0)
    assert res is not None, 'the sub-pixel path was not taken'
    assert [ents[i].kernel.decode() for i in range(cnt.value)] == ['upconv_subpixel_fwd<2 N tiles>' if (pn and cout == 64) else 'upconv_subpixel_fwd']
    y, scale, signs = res
    # the summed 2x2x2 weights are rounded to bf16 once instead of per tap: same tolerance class as the other kernels
    close(y, ref, dtype, 'sub-pixel vs oracle')
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)
    y2, scale2, signs2 = F.raw_conv(xg, wg, coef, False, True, bias=bg, act=True, slope=0.2, pixel_norm=pn,
                                    want_scale=True, want_signs=True)
    close(y, y2.double(), dtype, 'sub-pixel vs fused gather')
    if pn:
        np.testing.assert_allclose(scale.cpu().numpy(), scale2.cpu().numpy(), rtol=2e-2)
    diff = (signs ^ signs2).to(torch.int64) & 0xFFFFFFFF
    flipped = sum(int(((diff >> b_) & 1).sum()) for b_ in range(32))
    # sign bits can only differ where an activation is within rounding of zero (the two formulations round their weights
    # differently)
    assert flipped <= 2e-3 * signs.numel() * 32, flipped
    # and through the autograd wrapper: forward via the sub-pixel kernel, gradients via the gather kernels as before
    monkeypatch.setattr(F, '_NO_SUBPIXEL', False)
    xa, wa, ba = xg.clone().requires_grad_(True), wg.clone().requires_grad_(True), bg.clone().requires_grad_(True)
    ya = F.conv3d(xa, wa, coef, bias=ba, act=True, slope=0.2, pixel_norm=pn, upsample_in=True)
    close(ya, ref, dtype, 'autograd forward')
    gy = rnd(tuple(ya.shape), 44, dtype)
    gxa, gwa, gba = torch.autograd.grad(ya, [xa, wa, ba], cl(gy, dtype))
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)
    xb, wb, bb = xg.clone().requires_grad_(True), wg.clone().requires_grad_(True), bg.clone().requires_grad_(True)
    yb = F.conv3d(xb, wb, coef, bias=bb, act=True, slope=0.2, pixel_norm=pn, upsample_in=True)
    gxb, gwb, gbb = torch.autograd.grad(yb, [xb, wb, bb], cl(gy, dtype))
    for name, u, v in (('dx', gxa, gxb), ('dw', gwa, gwb), ('db', gba, gbb)):
        err = float(torch.linalg.vector_norm(u.double() - v.double()) / torch.linalg.vector_norm(v.double()))
        # the two forwards round their weights differently (per tap / per summed tap), so ~0.3 % of the LeakyReLU masks
        # differ between them, as between any two bf16 arithmetics: 3 % in relative L2 was seen
        assert err <= 5e-2, (name, err)


@pytest.mark.parametrize('case', [(2, 64, 32, (4, 8, 32)), (1, 128, 64, (2, 6, 64)), (3, 32, 32, (1, 2, 32))],
                         ids=['64to32', '128to64', '32to32_one_tile'])
def test_upconv_subpixel_weight_gradient(case, monkeypatch):
    """Weight and bias gradient of conv3d(upscale3d(x)) in sub-pixel form (csrc/subpix.hip: 64 (class, tap) tiles folded to
    the 27 taps) against the fp64 oracle's gradient of the 27-tap formulation, and against the fused-gather kernel."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp = case
    dtype = torch.bfloat16
    x = rnd((n, cin, *sp), 61, dtype)
    gy = rnd((n, cout, *[2 * v for v in sp]), 62, dtype)
    xr = x.clone()
    wr = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    yr = TF.conv3d(O.upscale3d(xr), wr.permute(4, 3, 0, 1, 2), padding=1)
    (gw,) = torch.autograd.grad(yr, wr, gy)
    refw, refb = gw * 0.37, gy.sum(dim=(0, 2, 3, 4))
    xg, gg = cl(x, dtype), cl(gy, dtype)
    lib = _lib.load()
    lib.sg_prof_enable(1)
    dw, db = F.raw_wgrad(xg, gg, (3, 3, 3), 0.37, ups=True, want_db=True)
    torch.cuda.synchronize()
    ents = (_lib.ProfEntry * 8)()
    cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 8, C.byref(cnt))
    lib.sg_prof_enable(0)
    assert [ents[i].kernel.decode() for i in range(cnt.value)] == ['upconv_subpixel_wgrad']
    np.testing.assert_allclose(dw.double().cpu().numpy(), refw.numpy(), rtol=2e-3, atol=2e-3 * float(refw.abs().max()))
    np.testing.assert_allclose(db.double().cpu().numpy(), refb.numpy(), rtol=2e-3, atol=2e-3 * float(refb.abs().max()))
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)
    dw2, db2 = F.raw_wgrad(xg, gg, (3, 3, 3), 0.37, ups=True, want_db=True)
    np.testing.assert_allclose(dw.cpu().numpy(), dw2.cpu().numpy(), rtol=1e-3, atol=1e-4 * float(dw2.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), db2.cpu().numpy(), rtol=1e-3, atol=1e-4 * float(db2.abs().max()))


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('shape', [(5, 48, 20), (16, 2048, 96)])   # the second: streamed-weight small-batch kernel
def test_dense(shape, dtype):
    from saragan_amd import functional as F
    n, cin, cout = shape
    x = rnd((n, cin), 8, dtype)
    w = rnd((cin, cout), 9, dtype)
    b = rnd((cout,), 10, torch.float32) * 0.5
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    xr, wr, br = x.clone().requires_grad_(True), wq.clone().requires_grad_(True), b.double().requires_grad_(True)
    yr = O.act(O.apply_bias(O.dense(xr, wr, 'leaky_relu', 0.2), br), 'leaky_relu', 0.2)
    gy = rnd(tuple(yr.shape), 10, dtype)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], gy)
    xg, wg = cl(x, dtype).requires_grad_(True), w.float().to(dev()).requires_grad_(True)
    bg = b.float().to(dev()).requires_grad_(True)
    yg = F.conv3d(xg, wg, coef, bias=bg, act=True, slope=0.2)
    close(yg.reshape(n, cout), yr, dtype)
    gxg, gwg, gbg = torch.autograd.grad(yg, [xg, wg, bg], cl(gy, dtype).reshape(yg.shape))
    close(gxg.reshape(n, cin), gxr, dtype)
    np.testing.assert_allclose(gwg.double().cpu().numpy(), gwr.numpy(), rtol=2e-3, atol=2e-3 * float(gwr.abs().max()))
    np.testing.assert_allclose(gbg.double().cpu().numpy(), gbr.numpy(), rtol=2e-3, atol=2e-3 * float(gbr.abs().max()))


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('c', [1, 8, 12, 32, 256, 4096])
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
    xr = x.clone().requires_grad_(True)
    yr = O.minibatch_stddev_layer(xr)
    gy = rnd(tuple(yr.shape), 16, dtype)
    (gxr,) = torch.autograd.grad(yr, [xr], gy)
    xg = cl(x, dtype).requires_grad_(True)
    yg = F.minibatch_stddev(xg)
    close(yg, yr, dtype, 'mbstd')
    (gxg,) = torch.autograd.grad(yg, [xg], cl(gy, dtype))
    close(gxg, gxr, dtype, 'mbstd backward')


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


def _mostly_close(got, ref, rt, at_scale, what, max_bad=1e-3):
    """assert_allclose for tensors behind a LeakyReLU mask: a handful of activations within rounding of zero take the
    other sign than the fp64 oracle's and disturb their 27-tap neighbourhood; at most `max_bad` of the elements may miss
    the tolerance, none by more than 20x."""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    scale = max(1e-30, float(ref.abs().max()))
    err = (got - ref).abs() - rt * ref.abs()
    bad = float((err > at_scale * scale).double().mean())
    assert bad <= max_bad, (what, bad)
    assert float(err.max()) <= 20 * at_scale * scale + 1e-12, (what, float(err.max()), scale)


def test_conv_epilogue_fused_downscale(monkeypatch):
    """downscale3d(leaky_relu(conv3d(x) + b)) with the pooling fused into the sliding-halo kernel's epilogue
    (sg_conv_epilogue.pool + sg_downscale_sum(1,2,1)) against O.downscale3d(O.act(...)): forward, first-order gradients
    and the gradient-penalty style second-order gradient (pgan/discriminator.py:33-44, loss.py:133-140)."""
    from saragan_amd import functional as F
    dtype = torch.bfloat16
    n, cin, cout, sp = 2, 32, 64, (6, 128, 256)
    x = rnd((n, cin, *sp), 61, dtype)
    w = rnd((3, 3, 3, cin, cout), 62, dtype)
    b = rnd((cout,), 63, torch.float32) * 0.3
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    br = b.double().clone().requires_grad_(True)
    yr = O.downscale3d(O.act(O.apply_bias(O.conv3d(xr, wr, 'leaky_relu', 0.2), br), 'leaky_relu', 0.2))
    xg = cl(x, dtype).requires_grad_(True)
    wg = w.float().to(dev()).requires_grad_(True)
    bg = b.float().to(dev()).requires_grad_(True)
    yg = F.conv3d_act_pool(xg, wg, coef, bg, 0.2)
    assert tuple(yg.shape) == (n, cout, 3, 64, 128)
    close(yg, yr, dtype, 'fused conv + bias + lrelu + downscale')
    # the unfused product path on the same inputs agrees too (same kernels, one more bf16 rounding of the full tensor)
    yu = F.downscale2x(F.conv3d(xg, wg, coef, bias=bg, act=True, slope=0.2), 0.125)
    close(yu, yr, dtype, 'unfused conv, downscale')
    gy = rnd(tuple(yr.shape), 64, dtype)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], gy, create_graph=True)
    gxg, gwg, gbg = torch.autograd.grad(yg, [xg, wg, bg], cl(gy, dtype), create_graph=True)
    _mostly_close(gxg, gxr, 1e-2, 1e-2, 'dx')
    _mostly_close(gwg, gwr, 2e-3, 4e-3, 'dw')
    _mostly_close(gbg, gbr, 2e-3, 4e-3, 'db')
    # second order: d/dw of sum(dx^2)
    (g2r,) = torch.autograd.grad((gxr * gxr).sum(), wr)
    (g2g,) = torch.autograd.grad((gxg.float() * gxg.float()).sum(), wg)
    _mostly_close(g2g, g2r, 2e-2, 2e-2, 'second-order dw', max_bad=5e-3)
    # requests no kernel fuses (f32 storage) are refused, not mis-computed
    assert F.raw_conv(xg.float(), wg, coef, False, bias=bg, act=True, want_signs=True, pool=True) is None
    # a backward nothing differentiates again never writes the up-scaled 64-channel gradient: the data gradient's two passes
    # and the weight gradient gather it from the pooled gradient and the sign words while they stage their tiles
    # (sg_conv_epilogue.in_mask_bits, sg_conv3d_wgrad_bias_up_masked): same gradients, bit for bit the same data gradient
    took_g = []
    real_g = F._pooled_backward_gather

    def spy_g(*a, **kw):
        r = real_g(*a, **kw)
        took_g.append(r is not None)
        return r
    monkeypatch.setattr(F, '_pooled_backward_gather', spy_g)
    yq = F.conv3d_act_pool(xg, wg, coef, bg, 0.2)
    gxq, gwq, gbq = torch.autograd.grad(yq, [xg, wg, bg], cl(gy, dtype))
    assert took_g == [True], 'the fused gather was not used'
    assert torch.equal(gxq, gxg.detach()), 'data gradient differs between the fused gather and the materialised gradient'
    _mostly_close(gwq, gwr, 2e-3, 4e-3, 'dw (gather)')
    _mostly_close(gbq, gbr, 2e-3, 4e-3, 'db (gather)')
    _mostly_close(gwq, gwg.detach(), 1e-3, 1e-3, 'dw (gather vs materialised)')
    _mostly_close(gbq, gbg.detach(), 1e-3, 1e-3, 'db (gather vs materialised)')
    monkeypatch.setattr(F, '_NO_GATHER_BWD', True)
    # without it: the up-scaled 64-channel gradient as two 32-channel tensors
    # (sg_upscale_nn_planes, sg_conv_epilogue.x_plane_channels): same gradients, bit for bit the same data gradient
    took = []
    real = F._pooled_backward_planes

    def spy(*a, **kw):
        r = real(*a, **kw)
        took.append(r is not None)
        return r
    monkeypatch.setattr(F, '_pooled_backward_planes', spy)
    yp = F.conv3d_act_pool(xg, wg, coef, bg, 0.2)
    gxp, gwp, gbp = torch.autograd.grad(yp, [xg, wg, bg], cl(gy, dtype))
    assert took == [True], 'the two-tensor layout was not used'
    # (round 4: the interleaved layouts run the one-pass kernel, the two-tensor layout still the two-pass K split: the same f32
    # sums in another order, so a few results round to the neighbouring bf16 value)
    nd = int((gxp.view(torch.int16) != gxg.detach().view(torch.int16)).sum())
    assert nd <= 1e-3 * gxp.numel(), ('data gradient differs between the layouts', nd)
    assert float((gxp.float() - gxg.detach().float()).abs().max()) <= 2.0 ** -7 * float(gxg.detach().float().abs().max())
    _mostly_close(gwp, gwr, 2e-3, 4e-3, 'dw (planes)')
    _mostly_close(gbp, gbr, 2e-3, 4e-3, 'db (planes)')
    # ... also when the layer's input is a LeakyReLU output whose mask this data gradient applies in its epilogue
    info = F.ActInfo(0.2)
    info.bits = F.sign_words(cl(rnd((n, cin, *sp), 65, dtype), dtype))
    info.consume(True)
    signs = F.raw_conv(xg.detach(), wg.detach(), coef, False, bias=bg.detach(), act=True, want_signs=True)[2]
    gyd = cl(gy, dtype)
    with torch.no_grad():
        res = real(gyd, xg.detach(), wg.detach(), signs, coef, 0.2, info, True, True, True)
        assert res is not None
        g_full = F._Up.apply(gyd, 0.125, signs, 0.2, (2, 2, 2))
        gx_ref = F._Conv.apply(g_full, wg.detach(), coef, True, False, None, info.bits, info.slope)
        gw_ref, gb_ref = F.raw_wgrad(xg.detach(), g_full, (3, 3, 3), coef, False, True)
    nd = int((res[0].view(torch.int16) != gx_ref.view(torch.int16)).sum())      # (K split vs one pass: f32 summation order)
    assert nd <= 1e-3 * gx_ref.numel(), ('masked data gradient differs between the layouts', nd)
    assert float((res[0].float() - gx_ref.float()).abs().max()) <= 2.0 ** -7 * float(gx_ref.float().abs().max())
    _mostly_close(res[1], gw_ref.reshape(res[1].shape), 1e-3, 1e-3, 'dw (planes vs one launch)')
    _mostly_close(res[2], gb_ref, 1e-3, 1e-3, 'db (planes vs one launch)')
    with torch.no_grad():
        resg = real_g(gyd, xg.detach(), wg.detach(), signs, coef, 0.2, info, True, True, True)
    assert resg is not None
    assert torch.equal(resg[0], gx_ref), 'masked data gradient differs between the fused gather and the materialised gradient'
    _mostly_close(resg[1], gw_ref.reshape(resg[1].shape), 1e-3, 1e-3, 'dw (gather vs one launch)')
    _mostly_close(resg[2], gb_ref, 1e-3, 1e-3, 'db (gather vs one launch)')


def test_pooled_backward_gather_equals_the_two_tensor_path_at_the_benchmarked_size(sg_env):
    """At the size the bench runs (batch 32, 32x128x128, the discriminator's 32 -> 64 layer): the fused masked gather
    (with the input's LeakyReLU mask in the epilogue) against the materialised gradient.
      * one-pass kernel (conv_fwd3p, round 4): the gather is BIT-IDENTICAL to the same kernel run on the materialised
        M * upscale3d(gy) / 8 (sg_upscale_nn writes exactly what the gather forms), and within f32 summation order
        (<= 1 bf16 ulp on <= 1e-3 of the elements) of the two-tensor path, which still runs the two-pass K split;
      * with the one-pass kernel switched off (SG_FWD_NO_3P=1) both paths run the K split and are bit-identical, as in
        round 3;
      * weight and bias gradient to f32 summation order."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    lib = _lib.load()
    n, d, h, w_ = 32, 32, 128, 128
    g_ = torch.Generator(device='cuda').manual_seed(92)

    def rn(shape):
        return torch.randn(shape, generator=g_, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last_3d)
    x, gy = rn((n, 32, d, h, w_)), rn((n, 64, d // 2, h // 2, w_ // 2))
    wt = torch.randn((3, 3, 3, 32, 64), generator=g_, device='cuda') * 0.05
    signs = F.sign_words(rn((n, 64, d, h, w_)))
    info = F.ActInfo(0.2)
    info.bits = F.sign_words(rn((n, 32, d, h, w_)))
    info.consume(True)
    with torch.no_grad():
        rp = F._pooled_backward_planes(gy, x, wt, signs, 0.05, 0.2, info, True, True, True)
        lib.sg_prof_enable(1)
        rg = F._pooled_backward_gather(gy, x, wt, signs, 0.05, 0.2, info, True, True, True)
        torch.cuda.synchronize()
        ents = (_lib.ProfEntry * 16)()
        cnt = C.c_int32(0)
        lib.sg_prof_collect(ents, 16, C.byref(cnt))
        lib.sg_prof_enable(0)
        names = [ents[i].kernel.decode() for i in range(cnt.value)]
        assert any('conv_fwd3p' in k for k in names), names
        g_full = F._Up.apply(gy, 0.125, signs, 0.2, (2, 2, 2))
        gx_same = F._Conv.apply(g_full, wt, 0.05, True, False, None, info.bits, info.slope)
        del g_full
    assert rp is not None and rg is not None
    assert torch.equal(gx_same, rg[0]), 'the fused gather differs from the same kernel on the materialised gradient'
    del gx_same
    ne = int((rp[0].view(torch.int16) != rg[0].view(torch.int16)).sum())
    assert ne <= 1e-3 * rg[0].numel(), ne
    assert float((rp[0].float() - rg[0].float()).abs().max() / rp[0].float().abs().max()) <= 2.0 ** -7
    assert float((rp[1] - rg[1]).abs().max() / rp[1].abs().max()) <= 1e-4
    assert float((rp[2] - rg[2]).abs().max() / rp[2].abs().max()) <= 1e-4
    sg_env(SG_FWD_NO_3P=1)
    with torch.no_grad():
        rg2 = F._pooled_backward_gather(gy, x, wt, signs, 0.05, 0.2, info, True, False, False)
    assert torch.equal(rp[0], rg2[0]), 'K split: data gradient differs between the fused gather and the two-tensor path'
    del rp, rg, rg2
    torch.cuda.empty_cache()


@pytest.mark.parametrize('n,sp', [(4, (6, 126, 256)), (2, (4, 252, 256)), (6, (10, 60, 288))])
def test_pooled_backward_gather_on_ragged_volumes(n, sp):
    """The fused masked gather of D's pooled backward (sg_conv_epilogue.in_mask_bits, sg_conv3d_wgrad_bias_up_masked) on
    volumes whose H is not a multiple of the 4-row tile and whose D gives an odd number of tile steps, with and without the
    input's own LeakyReLU mask in the epilogue: bit-identical data gradient, weight / bias gradient to f32 summation order,
    against the materialised masked up-scale followed by the plain kernels."""
    from saragan_amd import functional as F
    dtype = torch.bfloat16
    d, h, w_ = sp
    x = cl(rnd((n, 32, d, h, w_), 81, dtype), dtype)
    gy = cl(rnd((n, 64, d // 2, h // 2, w_ // 2), 82, dtype), dtype)
    wg = rnd((3, 3, 3, 32, 64), 83, torch.float32).to(dev())
    signs = F.sign_words(cl(rnd((n, 64, d, h, w_), 84, dtype), dtype))
    coef = 0.043
    for masked in (False, True):
        info = None
        if masked:
            info = F.ActInfo(0.2)
            info.bits = F.sign_words(cl(rnd((n, 32, d, h, w_), 85, dtype), dtype))
            info.consume(True)
        with torch.no_grad():
            res = F._pooled_backward_gather(gy, x, wg, signs, coef, 0.2, info, True, True, True)
            if res is None:
                pytest.fail('the library declined a shape the gather was written for')
            g_full = F._Up.apply(gy, 0.125, signs, 0.2, (2, 2, 2))
            if masked:
                gx_ref = F._Conv.apply(g_full, wg, coef, True, False, None, info.bits, info.slope)
            else:
                gx_ref = F._Conv.apply(g_full, wg, coef, True, False)
            gw_ref, gb_ref = F.raw_wgrad(x, g_full, (3, 3, 3), coef, False, True)
        assert torch.equal(res[0], gx_ref), ('data gradient', masked)
        _mostly_close(res[1], gw_ref.reshape(res[1].shape), 1e-3, 1e-3, 'dw (gather, ragged)')
        _mostly_close(res[2], gb_ref, 1e-3, 1e-3, 'db (gather, ragged)')


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('shape,factors', [((2, 40, 4, 6, 8), (2, 2, 2)), ((1, 8, 2, 4, 6), (1, 2, 2)), ((2, 33, 4, 4, 4), (2, 1, 2))])
def test_masked_block_sum_is_the_gradient_of_the_masked_upscale(shape, factors, dtype):
    """sg_downscale_sum_masked: gain * block sum of where(a < 0, slope, 1) * x in one pass (the gradient of
    upscale3d followed by a LeakyReLU mask, networks/ops.py:175-178 with :265-273), against the two-pass form and fp64;
    and as the backward of the masked up-scale Function."""
    from saragan_amd import functional as F
    n, c, d, h, w = shape
    x = rnd(shape, 81, dtype)
    a = rnd(shape, 82, dtype)
    xg, ag = cl(x, dtype), cl(a, dtype)
    bits = F.sign_words(ag)
    got = F._Down.apply(xg, 0.125, None, factors, bits, 0.2)
    m = torch.where(a.double() < 0, 0.2, 1.0)
    ref = (x.double() * m).reshape(n, c, d // factors[0], factors[0], h // factors[1], factors[1], w // factors[2], factors[2]).sum(dim=(3, 5, 7)) * 0.125
    close(got, ref, dtype, 'masked block sum')
    two = F._Down.apply(F._BiasActBwd.apply(xg, bits, 0.2, False)[0], 0.125, None, factors)
    close(got, two, dtype, 'one pass vs two passes')
    # as the backward of y = M * up(g)
    gsm = cl(rnd(tuple(ref.shape), 83, dtype), dtype).requires_grad_(True)
    up = F._Up.apply(gsm, 0.125, bits, 0.2, factors)
    (gg,) = torch.autograd.grad(up, gsm, xg)
    close(gg, ref, dtype, 'gradient of the masked up-scale')


@pytest.mark.parametrize('dtype', DT)
def test_premasked_double_backward_equals_separate_mask_passes(dtype, monkeypatch):
    """Gradient-penalty style second-order gradients through three conv + LeakyReLU layers wired the way the network
    code wires them (functional.ActInfo: each layer's mask applied by the next layer's data-gradient conv), once with
    the double backward's mask pull-backs fused into the consumers' conv epilogues (functional.BackInfo) and once as
    separate passes: same kernels otherwise, so the weight gradients agree to rounding of the intermediate tensors."""
    from saragan_amd import functional as F
    n, sp, chans = 2, (4, 8, 8), (8, 16, 16, 8)
    x0 = cl(rnd((n, chans[0], *sp), 91, dtype), dtype)
    ws = [rnd((3, 3, 3, chans[i], chans[i + 1]), 92 + i, torch.float32).to(dev()) for i in range(3)]
    bs = [(rnd((chans[i + 1],), 96 + i, torch.float32) * 0.2).to(dev()) for i in range(3)]

    passes = []
    real_bwd = F.raw_bias_act_bwd

    def counting(*a, **k):
        passes[-1] += 1
        return real_bwd(*a, **k)
    monkeypatch.setattr(F, 'raw_bias_act_bwd', counting)

    def second_order(no_premask):
        monkeypatch.setattr(F, '_NO_BACK_PREMASK', no_premask)
        passes.append(0)
        x = x0.clone().requires_grad_(True)
        wv = [w.clone().requires_grad_(True) for w in ws]
        bv = [b.clone().requires_grad_(True) for b in bs]
        h, info = x, None
        for i in range(3):
            out = F.ActInfo(0.2)
            if info is not None:
                info.consume(True)          # my data-gradient conv applies the previous layer's mask
            h = F.conv3d(h, wv[i], 0.1, bias=bv[i], act=True, slope=0.2, out_info=out, in_info=info)
            info = out
        info.consume(False)
        with F.skip_param_grads(wv + bv):
            (gx,) = torch.autograd.grad(h.float().sum(), x, create_graph=True)
        pen = (gx.float() ** 2).sum()
        return [g.float() for g in torch.autograd.grad(pen, wv)]

    fused = second_order(False)
    plain = second_order(True)
    assert passes[0] < passes[1], passes        # the fused run really skipped mask passes
    for i, (a, b) in enumerate(zip(fused, plain)):
        scale = float(b.abs().max())
        assert scale > 0
        tol = 2e-5 if dtype == torch.float32 else 2e-2       # bf16: the fused path skips a bf16 rounding of the masked tensor
        assert float((a - b).abs().max()) <= tol * scale, (i, float((a - b).abs().max()), scale)


def test_conv_64_to_32_split_over_input_channels(sg_env):
    """The 64 -> 32 channel 3x3x3 bf16 layers run as two sliding-halo passes over 32 input channels each, f32 partial
    sums in sg_conv_epilogue.workspace (sg_conv3d_fwd_workspace).  Same inputs through the split path, through the
    streamed single-pass kernel (SG_FWD_NO_KSPLIT=1) and through the oracle: bias, LeakyReLU, recorded sign words and the
    masked (second-order) epilogue; ragged H (a dead row in the last tile pair) and odd D."""
    import ctypes as C
    from saragan_amd import functional as F, _lib
    dtype = torch.bfloat16
    n, cin, cout, sp = 2, 64, 32, (5, 126, 256)     # 256 column pairs: one per block of the persistent grid
    x = rnd((n, cin, *sp), 71, dtype)
    w = rnd((3, 3, 3, cin, cout), 72, dtype)
    b = rnd((cout,), 73, torch.float32) * 0.3
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    yr = O.act(O.apply_bias(O.conv3d(x.double(), wq, 'leaky_relu', 0.2), b.double()), 'leaky_relu', 0.2)
    lin = O.apply_bias(O.conv3d(x.double(), wq, 'leaky_relu', 0.2), b.double())
    xg, wg, bg = cl(x, dtype), w.float().to(dev()), b.float().to(dev())
    lib = _lib.load()
    shp = _lib.ConvShape(n, *sp, cin, cout, 3, 3, 3, 0)
    assert lib.sg_conv3d_fwd_workspace(C.byref(shp), _lib.SG_BF16) == n * sp[0] * sp[1] * sp[2] * cout * 4

    def kernels_of(fn):
        lib.sg_prof_enable(1)
        out = fn()
        torch.cuda.synchronize()
        ents = (_lib.ProfEntry * 64)()
        cnt = C.c_int32(0)
        lib.sg_prof_collect(ents, 64, C.byref(cnt))
        lib.sg_prof_enable(0)
        return out, [ents[i].kernel.decode() for i in range(cnt.value)]

    (y3, s3), names3 = kernels_of(lambda: F.raw_conv(xg, wg, coef, False, bias=bg, act=True, want_signs=True)[::2])
    assert any('conv_fwd3p' in k for k in names3), names3        # round 4: one pass with sliding accumulators (conv3p.hip)
    sg_env(SG_FWD_NO_3P=1)
    (y2, s2), names = kernels_of(lambda: F.raw_conv(xg, wg, coef, False, bias=bg, act=True, want_signs=True)[::2])
    assert any('K split' in k for k in names), names
    sg_env(SG_FWD_NO_3P=0, SG_FWD_NO_KSPLIT=1)
    (y1, s1), names1 = kernels_of(lambda: F.raw_conv(xg, wg, coef, False, bias=bg, act=True, want_signs=True)[::2])
    assert not any('K split' in k or 'conv_fwd3p' in k for k in names1) and any('conv_fwd4' in k for k in names1), names1
    sg_env(SG_FWD_NO_KSPLIT=0)
    close(y3, yr, dtype, 'one-pass 64 -> 32')
    d3 = (y3.float() - y2.float()).abs()
    assert float((d3 > 0).float().mean()) < 2e-3 and float(d3.max()) <= 2.0 ** -6 * float(yr.abs().max()), (float(d3.max()),)
    close(y2, yr, dtype, 'split 64 -> 32')
    close(y1, yr, dtype, 'streamed 64 -> 32')
    # f32 partial sums in both: the two paths round the same f32 sums to bf16 (summation order differs by a few ulp of f32)
    d = (y2.float() - y1.float()).abs()
    assert float((d > 0).float().mean()) < 2e-3 and float(d.max()) <= 2.0 ** -6 * float(yr.abs().max()), (float(d.max()),)
    # sign words: equal except where the pre-activation is within rounding of zero
    flips = (s2 != s1)
    if bool(flips.any()):
        near0 = (lin.abs() < 1e-4 * float(lin.abs().max())).float().mean()
        assert float(flips.float().mean()) <= 64 * float(near0) + 1e-6, (float(flips.float().mean()), float(near0))
    # masked epilogue (the double-backward path): mask = recorded signs
    ym2 = F.raw_conv(xg, wg, coef, False, mask_bits=s1, mask_slope=0.2)[0]
    sg_env(SG_FWD_NO_KSPLIT=1)
    ym1 = F.raw_conv(xg, wg, coef, False, mask_bits=s1, mask_slope=0.2)[0]
    sg_env(SG_FWD_NO_KSPLIT=0)
    dm = (ym2.float() - ym1.float()).abs()
    assert float(dm.max()) <= 2.0 ** -6 * float(ym1.float().abs().max()), float(dm.max())


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('shape', [(2, 16, 3, 5, 7), (1, 6, 4, 4, 4), (2, 32, 1, 8, 8)])
def test_trilinear_up2x_and_adjoint(shape, dtype):
    """Trilinear x2 (half-pixel centres) against torch's CPU interpolate(mode='trilinear', align_corners=False) in fp64:
    forward, gradient (the adjoint kernel), second order, and down-sampling == the 2x2x2 mean."""
    import torch.nn.functional as TF
    from saragan_amd import functional as F
    x = rnd(shape, 71, dtype)
    xr = x.clone().requires_grad_(True)
    yr = TF.interpolate(xr, scale_factor=2, mode='trilinear', align_corners=False)
    xg = cl(x, dtype).requires_grad_(True)
    yg = F.upscale_trilinear2x(xg)
    close(yg, yr, dtype, 'trilinear up')
    gy = rnd(tuple(yr.shape), 72, dtype)
    (gxr,) = torch.autograd.grad(yr, xr, gy, create_graph=True)
    (gxg,) = torch.autograd.grad(yg, xg, cl(gy, dtype), create_graph=True)
    close(gxg, gxr, dtype, 'trilinear adjoint')
    assert not gxr.requires_grad            # linear op: the gradient does not depend on x
    gy2 = cl(gy, dtype).requires_grad_(True)
    gx2 = torch.autograd.grad(F.upscale_trilinear2x(xg), xg, gy2, create_graph=True)[0]
    (back,) = torch.autograd.grad(gx2, gy2, cl(x, dtype))       # adjoint of the adjoint = the forward op
    close(back, yr, dtype, 'adjoint of the adjoint')
    if all(s % 2 == 0 for s in shape[2:]):
        dn = TF.interpolate(x, scale_factor=0.5, mode='trilinear', align_corners=False)
        close(F.downscale2x(cl(x, dtype)), dn, dtype, 'trilinear down == 2x2x2 mean')


def test_upconv_data_gradient_pooled_in_the_conv_epilogue(monkeypatch):
    """Gradient of conv3d(upscale3d(x)) for x (pgan/generator.py:33-34): the block sum of the data gradient, pooled
    2 x 1 x 2 inside the sliding-halo kernel when the backward is not differentiated again, against the oracle's
    autograd and against the unfused product path (conv, then sg_downscale_sum)."""
    from saragan_amd import functional as F
    dtype = torch.bfloat16
    n, cin, cout, sp = 4, 64, 32, (4, 64, 128)           # fine volume 8 x 128 x 256: 2^20 voxels, 512 tile columns
    x = rnd((n, cin, *sp), 71, dtype)
    w = rnd((3, 3, 3, cin, cout), 72, dtype)
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    xr = x.clone().requires_grad_(True)
    yr = O.conv3d(O.upscale3d(xr), wq, 'leaky_relu', 0.2)
    gy = rnd(tuple(yr.shape), 73, dtype)
    (gxr,) = torch.autograd.grad(yr, [xr], gy)
    xg = cl(x, dtype).requires_grad_(True)
    wg = w.float().to(dev())
    calls = []
    real = F.raw_conv

    def spy(*a, **kw):
        calls.append(bool(kw.get('pool')))
        return real(*a, **kw)
    monkeypatch.setattr(F, 'raw_conv', spy)
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)      # (shapes the sub-pixel data gradient takes never get here: test_upconv_subpixel_data_gradient)
    yg = F.conv3d(xg, wg, coef, upsample_in=True)
    (gxg,) = torch.autograd.grad(yg, [xg], cl(gy, dtype))
    assert calls[-1] is True, 'the pooled epilogue was not used'
    _mostly_close(gxg, gxr, 1e-2, 1e-2, 'dx (pooled)')
    monkeypatch.setattr(F, '_NO_POOL_FUSION', True)
    yg2 = F.conv3d(xg, wg, coef, upsample_in=True)
    (gxu,) = torch.autograd.grad(yg2, [xg], cl(gy, dtype))
    assert calls[-1] is False
    _mostly_close(gxu, gxr, 1e-2, 1e-2, 'dx (unfused)')
    # differentiated again (create_graph): the unfused, differentiable path
    monkeypatch.setattr(F, '_NO_POOL_FUSION', False)
    yg3 = F.conv3d(xg, wg, coef, upsample_in=True)
    (gxd,) = torch.autograd.grad(yg3, [xg], cl(gy, dtype).requires_grad_(True), create_graph=True)
    assert calls[-1] is False and gxd.requires_grad
    _mostly_close(gxd, gxr, 1e-2, 1e-2, 'dx (create_graph)')


@pytest.mark.parametrize('dtype', DT)
def test_lerp_with_alpha_zero_or_one_prunes_the_faded_branch(dtype):
    """alpha = 0 (the stabilising half of a phase) / alpha = 1: networks.ops.lerp returns the live operand bit for bit
    without a pass over the tensors and without a graph edge to the other one (pgan/generator.py:100-101,
    pgan/discriminator.py:105); any other alpha goes through sg_axpby."""
    from saragan_amd import functional as F
    from saragan_amd.networks import ops
    a = cl(rnd((2, 8, 3, 4, 5), 81, dtype), dtype).requires_grad_(True)
    b = cl(rnd((2, 8, 3, 4, 5), 82, dtype), dtype).requires_grad_(True)
    g = cl(rnd((2, 8, 3, 4, 5), 83, dtype), dtype)
    for alpha, live, dead in ((0.0, b, a), (1.0, a, b)):
        out = ops.lerp(a * 1.0, b * 1.0, alpha)
        assert torch.equal(out, F.lerp(a, b, alpha, 1.0 - alpha)) and torch.equal(out, live)
        gl, gd = torch.autograd.grad(out, [live, dead], g, allow_unused=True)
        assert torch.equal(gl, g) and gd is None
    out2 = ops.lerp(a, b, 0.25)
    close(out2, 0.25 * a.double() + 0.75 * b.double(), dtype, 'lerp')
    ga, gb = torch.autograd.grad(out2, [a, b], g)
    close(ga, 0.25 * g.double(), dtype, 'd lerp / da')
    close(gb, 0.75 * g.double(), dtype, 'd lerp / db')


def test_conv_epilogue_fused_downscale_hw_pairs():
    """The same fusion for layers with more than 32 input channels (pgan/discriminator.py:33-44 at the lower
    resolutions): the streamed kernel pools H x W pairs (sg_conv_epilogue.pool = 2), sg_downscale_sum(2,1,1) the D pairs."""
    from saragan_amd import functional as F
    dtype = torch.bfloat16
    n, cin, cout, sp = 2, 48, 64, (8, 128, 128)
    x = rnd((n, cin, *sp), 91, dtype)
    w = rnd((3, 3, 3, cin, cout), 92, dtype)
    b = rnd((cout,), 93, torch.float32) * 0.3
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    br = b.double().clone().requires_grad_(True)
    yr = O.downscale3d(O.act(O.apply_bias(O.conv3d(xr, wr, 'leaky_relu', 0.2), br), 'leaky_relu', 0.2))
    xg = cl(x, dtype).requires_grad_(True)
    wg = w.float().to(dev()).requires_grad_(True)
    bg = b.float().to(dev()).requires_grad_(True)
    assert F._pool_mode(xg, w.shape[:3], cin, cout) == 2
    res = F.raw_conv(xg.detach(), wg.detach(), coef, False, bias=bg.detach(), act=True, want_signs=True, pool=2)
    assert res is not None and tuple(res[0].shape) == (n, cout, 8, 64, 64), 'the H x W pooled epilogue did not engage'
    full, _, signs_full = F.raw_conv(xg.detach(), wg.detach(), coef, False, bias=bg.detach(), act=True, want_signs=True)
    assert torch.equal(res[2], signs_full), 'sign words of the pooled launch differ from the plain one'
    yg = F.conv3d_act_pool(xg, wg, coef, bg, 0.2)
    assert tuple(yg.shape) == (n, cout, 4, 64, 64)
    close(yg, yr, dtype, 'fused conv + bias + lrelu + downscale (H x W pairs)')
    gy = rnd(tuple(yr.shape), 94, dtype)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], gy)
    gxg, gwg, gbg = torch.autograd.grad(yg, [xg, wg, bg], cl(gy, dtype))
    _mostly_close(gxg, gxr, 1e-2, 1e-2, 'dx')
    _mostly_close(gwg, gwr, 2e-3, 4e-3, 'dw')
    _mostly_close(gbg, gbr, 2e-3, 4e-3, 'db')
    # ragged extent in W (even, not a multiple of 32) and odd D: still whole H x W blocks
    x2 = cl(rnd((4, 64, 5, 64, 96), 95, dtype), dtype)
    w2 = rnd((3, 3, 3, 64, 128), 96, dtype).float().to(dev())
    r2 = F.raw_conv(x2, w2, 0.05, False, pool=2)
    assert r2 is not None
    f2 = F.raw_conv(x2, w2, 0.05, False)[0].float()
    ref2 = torch.nn.functional.avg_pool3d(f2, (1, 2, 2))
    assert float((r2[0].float() - ref2).abs().max()) <= 1e-2 * float(ref2.abs().max())


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('ups', [False, True])
def test_generator_tail_with_to_rgb_as_one_node(dtype, ups):
    """y = pixel_norm(leaky_relu(conv3d(x) + b)), img = to_rgb(y) (pgan/generator.py:33-45,96-97) as one autograd node:
    when only img is used, to_rgb's full-resolution data gradient is formed inside the pixel-norm backward
    (sg_pixel_norm_act_bwd_pw); when y has another consumer, it is a tensor as before.  Both against the oracle."""
    from saragan_amd import functional as F
    n, cin, c, sp = 2, 16, 32, (3, 6, 8)
    x = rnd((n, cin, *sp), 101, dtype)
    w = rnd((3, 3, 3, cin, c), 102, dtype)
    b = rnd((c,), 103, torch.float32) * 0.3
    wr = rnd((1, 1, 1, c, 1), 104, dtype)
    br = rnd((1,), 105, torch.float32) * 0.3
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    coef_r = O.runtime_coef(wr.shape, 'linear', None)
    wq = (w * coef).to(dtype).double() / coef
    wrq = (wr * coef_r).to(dtype).double() / coef_r
    leaves = [t.clone().requires_grad_(True) for t in (x, wq, b.double(), wrq, br.double())]
    xr, w_r, b_r, wr_r, br_r = leaves
    xin = O.upscale3d(xr) if ups else xr
    y_r = O.pixel_norm(O.act(O.apply_bias(O.conv3d(xin, w_r, 'leaky_relu', 0.2), b_r), 'leaky_relu', 0.2))
    img_r = O.apply_bias(O.conv3d(y_r, wr_r, 'linear', None), br_r)
    gl = [cl(x, dtype).requires_grad_(True)] + [t.float().to(dev()).requires_grad_(True) for t in (w, b, wr, br)]
    xg, wg, bg, wrg, brg = gl
    img, y = F.conv3d_pn_to_rgb(xg, wg, coef, bg, ups, 0.2, 1e-8, None, wrg, coef_r, brg)
    # bf16: y is rounded before to_rgb reads it
    close(y, y_r, dtype, 'y')
    close(img, img_r, dtype, 'img')
    g_img = rnd(tuple(img_r.shape), 106, dtype)
    ref = torch.autograd.grad(img_r, leaves, g_img, retain_graph=True)
    got = torch.autograd.grad(img, gl, cl(g_img, dtype), retain_graph=True)
    rt, at = (2e-3, 2e-3) if dtype == torch.float32 else (2e-2, 2e-2)
    for name, a, r in zip(('dx', 'dw', 'db', 'dw_rgb', 'db_rgb'), got, ref):
        _mostly_close(a, r, rt, at, name + ' (fused)')
    # y consumed elsewhere too: gradients of img and y add up
    g_y = rnd(tuple(y_r.shape), 107, dtype)
    ref2 = torch.autograd.grad([img_r, y_r], leaves, [g_img, g_y])
    got2 = torch.autograd.grad([img, y], gl, [cl(g_img, dtype), cl(g_y, dtype)])
    for name, a, r in zip(('dx', 'dw', 'db', 'dw_rgb', 'db_rgb'), got2, ref2):
        _mostly_close(a, r, rt, at, name + ' (y also used)')


def test_generator_uses_the_fused_tail(monkeypatch):
    """networks.ops: to_rgb of an unmaterialised generator stage becomes the one-node tail; a stage that another
    consumer materialised first falls back to the plain pointwise convolution."""
    from saragan_amd import functional as F
    from saragan_amd import varstore
    from saragan_amd.networks import ops
    calls = []
    real = F.conv3d_pn_to_rgb
    monkeypatch.setattr(F, 'conv3d_pn_to_rgb', lambda *a: (calls.append(1), real(*a))[1])
    store = varstore.VariableStore(dev(), seed=3)
    monkeypatch.setitem(varstore.COMPUTE_DTYPE, 'dtype', torch.bfloat16)
    with varstore.use_store(store):
        x = cl(rnd((2, 16, 2, 8, 8), 111, torch.bfloat16), torch.bfloat16)
        with varstore.variable_scope('g'):
            with varstore.variable_scope('c1'):
                h = ops.pixel_norm(ops.act(ops.apply_bias(ops.conv3d(x, 32, (3, 3, 3), 'leaky_relu', 0.2)), 'leaky_relu', 0.2))
            with varstore.variable_scope('rgb'):
                img = ops.materialize(ops.to_rgb(h, 1))
            assert calls == [1] and tuple(img.shape) == (2, 1, 2, 8, 8)
            # the stage's handle now holds its value: a second consumer reuses it
            again = ops.materialize(h)
            with varstore.variable_scope('rgb', reuse=True):
                img2 = ops.materialize(ops.to_rgb(h, 1))
            assert calls == [1] and tuple(again.shape) == (2, 32, 2, 8, 8)
            assert torch.equal(img, img2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('cin', [1, 3])
def test_from_rgb_backward_in_one_pass(dtype, cin, monkeypatch):
    """from_rgb (pointwise conv from 1 / 3 channels + bias + LeakyReLU, pgan/discriminator.py:9-12): when nothing
    differentiates the backward again, data, weight and bias gradients come from ONE pass over the output gradient
    (sg_conv3d_pw_bwd); against the oracle and against the separate kernels."""
    from saragan_amd import functional as F
    n, c, sp = 3, 32, (3, 6, 9)
    x = rnd((n, cin, *sp), 121, dtype)
    w = rnd((1, 1, 1, cin, c), 122, dtype)
    b = rnd((c,), 123, torch.float32) * 0.3
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wq = (w * coef).to(dtype).double() / coef
    leaves = [t.clone().requires_grad_(True) for t in (x, wq, b.double())]
    y_r = O.act(O.apply_bias(O.conv3d(leaves[0], leaves[1], 'leaky_relu', 0.2), leaves[2]), 'leaky_relu', 0.2)
    gy = rnd(tuple(y_r.shape), 124, dtype)
    ref = torch.autograd.grad(y_r, leaves, gy)
    used = []
    real = F._pw_backward
    monkeypatch.setattr(F, '_pw_backward', lambda *a: (lambda r: (used.append(r is not None), r)[1])(real(*a)))
    gl = [cl(x, dtype).requires_grad_(True), w.float().to(dev()).requires_grad_(True), b.float().to(dev()).requires_grad_(True)]
    y = F.conv3d(gl[0], gl[1], coef, bias=gl[2], act=True, slope=0.2)
    close(y, y_r, dtype, 'from_rgb')
    got = torch.autograd.grad(y, gl, cl(gy, dtype), retain_graph=True)
    assert used == [True], 'the one-pass backward was not used'
    rt, at = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 1e-2)
    for name, a, r in zip(('dx', 'dw', 'db'), got, ref):
        _mostly_close(a, r, rt, at, name)
    monkeypatch.setattr(F, '_NO_RGB_FUSION', True)
    sep = torch.autograd.grad(y, gl, cl(gy, dtype))
    assert used == [True]
    for name, a, r in zip(('dx', 'dw', 'db'), got, sep):
        _mostly_close(a, r, rt, at, name + ' vs separate kernels')


def test_pixel_norm_backward_in_the_data_gradient_epilogue():
    """The data-gradient conv whose result is the gradient for a pixel-norm stage's output applies that stage's backward
    (pixel norm + LeakyReLU mask) in its epilogue (sg_conv_epilogue.pn_bwd_y): against the two separate passes, conv then
    sg_pixel_norm_act_bwd, which round the intermediate to bf16 once more, and against the oracle's arithmetic."""
    from saragan_amd import functional as F
    dtype = torch.bfloat16
    n, c, sp = 2, 32, (6, 128, 256)
    g = cl(rnd((n, c, *sp), 131, dtype), dtype)                    # gradient arriving at the NEXT conv's output
    w = rnd((3, 3, 3, c, c), 132, dtype).float().to(dev())          # the next conv's weights (its data gradient: flip)
    y = cl(rnd((n, c, *sp), 133, dtype), dtype)                    # the stage's output
    scale = (torch.rand(n * sp[0] * sp[1] * sp[2], generator=torch.Generator().manual_seed(134)) + 0.5).to(dev())
    signs = F.sign_words(cl(rnd((n, c, *sp), 135, dtype), dtype))
    coef = 0.05
    fused = F.raw_conv(g, w, coef, True, mask_bits=signs, mask_slope=0.2, pn_bwd=(y, scale))
    assert fused is not None, 'the epilogue did not engage'
    gy = F.raw_conv(g, w, coef, True)[0]
    two, _ = F._PnActBwd.apply(gy, y, scale, signs, 0.2, False)
    ref = gy.double()
    yd = y.double()
    exact = scale.double().view(n, 1, *sp) * (ref - yd * (ref * yd).mean(dim=1, keepdim=True))
    exact = torch.where(F_bits(signs, n, c, sp), exact * 0.2, exact)
    err_f = float((fused[0].double() - exact).abs().max() / exact.abs().max())
    err_t = float((two.double() - exact).abs().max() / exact.abs().max())
    assert err_f <= 1.5e-2 and err_t <= 1.5e-2, (err_f, err_t)
    # refused where no kernel has the epilogue (f32 storage), not mis-computed
    assert F.raw_conv(g.float(), w, coef, True, mask_bits=signs, mask_slope=0.2, pn_bwd=(y.float(), scale)) is None


def F_bits(words, n, c, sp):
    """bool [n,c,d,h,w]: bit (ch % 32) of sign word [n,d,h,w,ch // 32]."""
    wv = words.view(n, *sp, (c + 31) // 32).long() & 0xFFFFFFFF
    ch = torch.arange(c, device=words.device)
    bits = (wv[..., ch // 32] >> (ch % 32)) & 1
    return bits.permute(0, 4, 1, 2, 3).bool()


def test_two_generator_stages_pixel_norm_backward_fused_into_the_data_gradient(monkeypatch):
    """Two stacked generator stages through networks.ops (conv -> bias -> LeakyReLU -> pixel_norm, twice,
    pgan/generator.py:33-45): the second conv registers for the first stage's backward and applies it in the epilogue of
    its data gradient (sg_conv_epilogue.pn_bwd_y); the first stage skips its own pass and takes the bias gradient from
    the weight-gradient kernel.  Every gradient against the oracle; and the same graph with the epilogue disabled."""
    from saragan_amd import functional as F
    from saragan_amd import varstore
    from saragan_amd.networks import ops
    dtype = torch.bfloat16
    n, cin, c, sp = 2, 16, 32, (4, 128, 256)
    x = rnd((n, cin, *sp), 141, dtype)
    monkeypatch.setitem(varstore.COMPUTE_DTYPE, 'dtype', dtype)
    used = []
    real = F.raw_conv

    def spy(*a, **kw):
        r = real(*a, **kw)
        if kw.get('pn_bwd') is not None:
            used.append(r is not None)
        return r
    monkeypatch.setattr(F, 'raw_conv', spy)

    def run(store):
        xg = cl(x, dtype).requires_grad_(True)
        with varstore.use_store(store), varstore.variable_scope('g'):
            h = xg
            for name in ('c1', 'c2'):
                with varstore.variable_scope(name):
                    h = ops.pixel_norm(ops.act(ops.apply_bias(ops.conv3d(h, c, (3, 3, 3), 'leaky_relu', 0.2)), 'leaky_relu', 0.2))
            y = ops.materialize(h)
        names = ['g/c1/weight', 'g/c1/bias', 'g/c2/weight', 'g/c2/bias']
        leaves = [xg] + [store.vars[k] for k in names]
        return y, leaves, names

    store = varstore.VariableStore(dev(), seed=5)
    y, leaves, names = run(store)
    gy = rnd(tuple(y.shape), 142, dtype)
    got = torch.autograd.grad(y, leaves, cl(gy, dtype))
    assert used == [True], 'the pixel-norm backward epilogue was not used'
    # oracle on the same (rounded) weights
    ws = {k: store.vars[k].detach().double().cpu() for k in names}
    lr = [x.clone().requires_grad_(True)]
    hr = lr[0]
    for nm in ('c1', 'c2'):
        w = ws[f'g/{nm}/weight']
        coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
        wq = ((w * coef).to(dtype).double() / coef).requires_grad_(True)
        b = ws[f'g/{nm}/bias'].clone().requires_grad_(True)
        lr += [wq, b]
        hr = O.pixel_norm(O.act(O.apply_bias(O.conv3d(hr, wq, 'leaky_relu', 0.2), b), 'leaky_relu', 0.2))
    close(y, hr, dtype, 'two stages')
    ref = torch.autograd.grad(hr, lr, gy)
    for name, a, r in zip(['dx'] + names, got, ref):
        _mostly_close(a, r, 3e-2, 3e-2, name, max_bad=5e-3)
    # the same graph with the stage's own pass: same gradients to bf16 rounding of one intermediate
    monkeypatch.setattr(F, '_NO_PN_EPILOGUE', True)
    y2, leaves2, _ = run(store)
    sep = torch.autograd.grad(y2, leaves2, cl(gy, dtype))
    assert used == [True]
    for name, a, r in zip(['dx'] + names, got, sep):
        _mostly_close(a, r, 3e-2, 3e-2, name + ' vs separate pass', max_bad=5e-3)


@pytest.mark.parametrize('rgb_first', [False, True])
def test_pixel_norm_stage_with_a_conv_and_a_to_rgb_consumer(monkeypatch, rgb_first):
    """A pixel-norm stage read by the next 3x3x3 conv FIRST (which signs up for the stage's backward) and by to_rgb
    afterwards (an ordinary consumer, as during a mixing phase): nobody may skip the stage's own backward then.
    Gradients against the oracle."""
    from saragan_amd import varstore
    from saragan_amd.networks import ops
    dtype = torch.bfloat16
    n, cin, c, sp = 2, 16, 32, (4, 128, 256)
    x = rnd((n, cin, *sp), 151, dtype)
    monkeypatch.setitem(varstore.COMPUTE_DTYPE, 'dtype', dtype)
    store = varstore.VariableStore(dev(), seed=6)
    xg = cl(x, dtype).requires_grad_(True)
    with varstore.use_store(store), varstore.variable_scope('g'):
        with varstore.variable_scope('c1'):
            h1 = ops.pixel_norm(ops.act(ops.apply_bias(ops.conv3d(xg, c, (3, 3, 3), 'leaky_relu', 0.2)), 'leaky_relu', 0.2))
        if rgb_first:       # to_rgb meets the unmaterialised stage: one node for both; the conv reads its second output
            with varstore.variable_scope('rgb'):
                img = ops.materialize(ops.to_rgb(h1, 1))
        with varstore.variable_scope('c2'):
            h2 = ops.materialize(ops.apply_bias(ops.conv3d(h1, c, (3, 3, 3), 'linear')))
        if not rgb_first:
            with varstore.variable_scope('rgb'):
                img = ops.materialize(ops.to_rgb(h1, 1))
    names = ['g/c1/weight', 'g/c1/bias', 'g/c2/weight', 'g/c2/bias', 'g/rgb/weight', 'g/rgb/bias']
    leaves = [xg] + [store.vars[k] for k in names]
    g2, gi = rnd(tuple(h2.shape), 152, dtype), rnd(tuple(img.shape), 153, dtype)
    got = torch.autograd.grad([h2, img], leaves, [cl(g2, dtype), cl(gi, dtype)])
    ws = {k: store.vars[k].detach().double().cpu() for k in names}

    def q(w, act):
        coef = O.runtime_coef(w.shape, act, 0.2 if act == 'leaky_relu' else None)
        return ((w * coef).to(dtype).double() / coef).requires_grad_(True)
    lr = [x.clone().requires_grad_(True), q(ws[names[0]], 'leaky_relu'), ws[names[1]].clone().requires_grad_(True),
          q(ws[names[2]], 'linear'), ws[names[3]].clone().requires_grad_(True), q(ws[names[4]], 'linear'),
          ws[names[5]].clone().requires_grad_(True)]
    r1 = O.pixel_norm(O.act(O.apply_bias(O.conv3d(lr[0], lr[1], 'leaky_relu', 0.2), lr[2]), 'leaky_relu', 0.2))
    r2 = O.apply_bias(O.conv3d(r1, lr[3], 'linear', None), lr[4])
    ri = O.apply_bias(O.conv3d(r1, lr[5], 'linear', None), lr[6])
    ref = torch.autograd.grad([r2, ri], lr, [g2, gi])
    for name, a, r in zip(['dx'] + names, got, ref):
        _mostly_close(a, r, 3e-2, 3e-2, name, max_bad=5e-3)


W16_CASES = [
    # n, cin, cout, (d, h, w), ups, with bias gradient
    (8, 128, 128, (4, 16, 16), False, True),     # the 4 x 16 x 16 level of the 'm' network
    (32, 64, 64, (4, 16, 16), False, False),     # few columns: the grid shrinks to keep two columns per block
    (8, 256, 64, (4, 16, 16), True, True),       # conv3d(upscale3d(x)): x is the 2 x 8 x 8 level
    (16, 40, 72, (6, 24, 16), False, True),      # ragged: channels not multiples of 32, H = 3 tiles, D = 3 slides
    (4, 32, 32, (2, 8, 16), False, True),        # a single D tile per column: stays on the tap-per-wave kernel
    # small batches (round 5): fewer columns than blocks -- wave groups and whole blocks without a column idle and send nothing
    (2, 128, 128, (4, 16, 16), False, True),
    (1, 128, 512, (4, 16, 16), False, True),
    (2, 512, 128, (4, 16, 16), True, True),
    (3, 64, 64, (4, 16, 16), False, False),
]


@pytest.mark.parametrize('case', W16_CASES, ids=[f'n{c[0]}_{c[1]}to{c[2]}at{"x".join(map(str, c[3]))}{"ups" if c[4] else ""}' for c in W16_CASES])
def test_wgrad_16_wide_levels_on_the_sliding_halo_kernel(case, sg_env, monkeypatch):
    """conv_wgrad3l in tiles of 2 x 8 x 16 voxels (the 4 x 16 x 16 levels; discriminator.py:48-68, generator.py:26-45 at
    phase 3): against the fp64 oracle, and within f32 rounding (the sums run in another order) of the tap-per-wave
    kernel it replaces (SG_WGRAD_NO_W16=1)."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp, ups, want_db = case
    dtype = torch.bfloat16
    xs = tuple(v // 2 for v in sp) if ups else sp
    x = rnd((n, cin, *xs), 301, dtype)
    gy = rnd((n, cout, *sp), 302, dtype)
    xr = x.clone()
    wr = torch.zeros((3, 3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    yr = O.conv3d(O.upscale3d(xr) if ups else xr, wr, 'linear', None)
    coef = O.runtime_coef(wr.shape, 'linear', None)
    (gwr,) = torch.autograd.grad(yr, [wr], gy)
    lib = _lib.load()

    def run():
        lib.sg_prof_enable(1)
        dw, db = F.raw_wgrad(cl(x, dtype), cl(gy, dtype), (3, 3, 3), coef, ups=ups, want_db=want_db)
        torch.cuda.synchronize()
        ents = (_lib.ProfEntry * 8)()
        cnt = C.c_int32(0)
        lib.sg_prof_collect(ents, 8, C.byref(cnt))
        lib.sg_prof_enable(0)
        return dw, db, [ents[i].kernel.decode() for i in range(cnt.value)]

    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)   # (a sub-pixel weight gradient would take an up-sampled layer first)
    dw, db, names = run()
    sg_env(SG_WGRAD_NO_W16=1)
    dw0, db0, names0 = run()
    expect = 'conv_wgrad3l<ups,w16>' if ups else 'conv_wgrad3l<w16>'
    if sp[0] >= 4:
        assert names == [expect], names
    else:
        assert names != [expect]
    assert names0 != [expect]
    ref = gwr.numpy()
    np.testing.assert_allclose(dw.double().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max(), err_msg='dw vs oracle')
    np.testing.assert_allclose(dw.cpu().numpy(), dw0.cpu().numpy(), rtol=1e-4, atol=1e-5 * np.abs(ref).max(), err_msg='dw vs wgrad2')
    if want_db:
        dbr = gy.sum(dim=(0, 2, 3, 4)).numpy()
        np.testing.assert_allclose(db.double().cpu().numpy(), dbr, rtol=1e-4, atol=1e-4 * np.abs(dbr).max(), err_msg='db')
        np.testing.assert_allclose(db.cpu().numpy(), db0.cpu().numpy(), rtol=1e-4, atol=1e-5 * np.abs(dbr).max())


@pytest.mark.parametrize('case', [(2, 32, 32, (4, 16, 32), False), (2, 32, 64, (4, 128, 256), True), (4, 64, 64, (1, 8, 8), False), (2, 1, 16, (4, 8, 8), None)])
def test_wgrad_accumulate_equals_the_add_of_the_finished_gradient(case):
    """sg_conv3d_wgrad_bias_ex.  SG_WGRAD_ACCUMULATE (a second contribution to a parameter's gradient, optimization.py:128-163
    over networks/loss.py:136-140): dw += coef * sum, bit for bit what adding the finished gradient gives; the bias gradient is
    written.  SG_WGRAD_CLEAN_WORKSPACE: a kept workspace that is zero on entry is zero again afterwards, no memset in between.
    Both declined (SG_EUNSUPPORTED, nothing touched) on the pointwise path."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    n, cin, cout, sp, masked = case
    dtype = torch.bfloat16
    lib = _lib.load()
    k = (1, 1, 1) if masked is None else ((1, 3, 3) if sp[0] == 1 else (3, 3, 3))
    x = cl(rnd((n, cin, *sp), 401, dtype), dtype)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shp = _lib.ConvShape(n, *sp, cin, cout, *k, 0)
    ws_bytes = lib.sg_conv3d_wgrad_workspace(C.byref(shp), _lib.SG_BF16)
    ws = torch.empty(ws_bytes, device=dev(), dtype=torch.uint8)
    first = torch.randn((*k, cin, cout), device=dev())
    if masked:
        half = tuple(v // 2 for v in sp)
        gy = cl(rnd((n, cout, *half), 402, dtype), dtype)
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * sp[0] * sp[1] * sp[2], cout // 32), device=dev(), dtype=torch.int32)
        ref = torch.empty_like(first)
        _lib.check(lib.sg_conv3d_wgrad_bias_up_masked(x.data_ptr(), gy.data_ptr(), bits.data_ptr(), 0.2, 0.125, ref.data_ptr(), None, 0.05,
                                                      ws.data_ptr(), ws_bytes, C.byref(shp), _lib.SG_BF16, st))
        margs = (bits.data_ptr(), 0.2, 0.125)
    else:
        gy = cl(rnd((n, cout, *sp), 402, dtype), dtype)
        ref, _ = F.raw_wgrad(x, gy, k, 0.05)
        margs = (None, 0.0, 1.0)
    acc = first.clone()
    db = torch.full((cout,), 7.0, device=dev())
    rc = lib.sg_conv3d_wgrad_bias_ex(x.data_ptr(), gy.data_ptr(), *margs, acc.data_ptr(), None if masked else db.data_ptr(), 0.05,
                                     _lib.SG_WGRAD_ACCUMULATE, ws.data_ptr(), ws_bytes, C.byref(shp), _lib.SG_BF16, st)
    kept = torch.zeros(ws_bytes, device=dev(), dtype=torch.uint8)
    if masked is None:
        assert rc == _lib.SG_EUNSUPPORTED and torch.equal(acc, first)
        rc = lib.sg_conv3d_wgrad_bias_ex(x.data_ptr(), gy.data_ptr(), *margs, acc.data_ptr(), None, 0.05, _lib.SG_WGRAD_CLEAN_WORKSPACE,
                                         kept.data_ptr(), ws_bytes, C.byref(shp), _lib.SG_BF16, st)
        assert rc == _lib.SG_EUNSUPPORTED and torch.equal(acc, first) and int(kept.count_nonzero()) == 0
        return
    _lib.check(rc)
    # the kept workspace: two calls in a row without a memset, same result as the plain entry point, clean afterwards
    clean_bytes = lib.sg_conv3d_wgrad_clean_bytes(C.byref(shp), _lib.SG_BF16)
    assert 0 < clean_bytes <= ws_bytes
    for _ in range(2):
        got = torch.empty_like(first)
        gdb = torch.empty(cout, device=dev())
        _lib.check(lib.sg_conv3d_wgrad_bias_ex(x.data_ptr(), gy.data_ptr(), *margs, got.data_ptr(), gdb.data_ptr(), 0.05,
                                               _lib.SG_WGRAD_CLEAN_WORKSPACE, kept.data_ptr(), ws_bytes, C.byref(shp), _lib.SG_BF16, st))
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
        assert int(kept[:clean_bytes].count_nonzero()) == 0
    import saragan_amd
    if not masked:      # atomics: the sums themselves differ from launch to launch in the last bits unless reproducible
        np.testing.assert_allclose(acc.cpu().numpy(), (first + ref).cpu().numpy(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
        dbr = gy.float().sum(dim=(0, 2, 3, 4))
        np.testing.assert_allclose(db.cpu().numpy(), dbr.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(dbr.abs().max()))
    saragan_amd.set_deterministic(True)
    try:
        ws2 = torch.empty(lib.sg_conv3d_wgrad_workspace(C.byref(shp), _lib.SG_BF16), device=dev(), dtype=torch.uint8)
        ref2 = torch.empty_like(first)
        acc2 = first.clone()
        if masked:
            _lib.check(lib.sg_conv3d_wgrad_bias_up_masked(x.data_ptr(), gy.data_ptr(), bits.data_ptr(), 0.2, 0.125, ref2.data_ptr(), None,
                                                          0.05, ws2.data_ptr(), ws2.numel(), C.byref(shp), _lib.SG_BF16, st))
        else:
            _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), gy.data_ptr(), ref2.data_ptr(), None, 0.05, ws2.data_ptr(), ws2.numel(),
                                                C.byref(shp), _lib.SG_BF16, st))
        _lib.check(lib.sg_conv3d_wgrad_bias_ex(x.data_ptr(), gy.data_ptr(), *margs, acc2.data_ptr(), None, 0.05, _lib.SG_WGRAD_ACCUMULATE,
                                               ws2.data_ptr(), ws2.numel(), C.byref(shp), _lib.SG_BF16, st))
        assert torch.equal(acc2, first + ref2)
    finally:
        saragan_amd.set_deterministic(False)


def test_kept_wgrad_workspace_is_shared_by_layers_of_any_channel_count(monkeypatch):
    """SG_WGRAD_CLEAN_WORKSPACE through functional.raw_wgrad: layers whose workspaces have the same size share one kept buffer.
    A 16-channel layer (partial 32 x 32 tiles, the generic kernel, bias gradient by the fallback pass) must leave it clean for a
    32-channel layer that follows (sliding-halo kernel with the bias gradient in its spare tap) -- same results as with a fresh
    workspace and a memset per call."""
    from saragan_amd import functional as F
    dtype = torch.bfloat16
    shapes = [(2, 16, 32, (4, 8, 8)), (2, 32, 32, (5, 128, 256)), (2, 24, 8, (3, 5, 7)), (2, 32, 32, (4, 16, 32))]
    data = [(cl(rnd((n, ci, *sp), 500 + i, dtype), dtype), cl(rnd((n, co, *sp), 600 + i, dtype), dtype)) for i, (n, ci, co, sp) in enumerate(shapes)]

    def sweep():
        out = []
        for _ in range(2):
            for x, gy in data:
                out.append(F.raw_wgrad(x, gy, (3, 3, 3), 0.05, False, want_db=True))
        torch.cuda.synchronize()
        return out

    F._CLEAN_WS.clear()
    F._CLEAN_WS_DECLINED.clear()
    kept = sweep()
    assert len(F._CLEAN_WS) >= 1 and not F._CLEAN_WS_DECLINED
    monkeypatch.setattr(F, '_NO_CLEAN_WS', True)
    fresh = sweep()
    for (dw, db), (dw0, db0) in zip(kept, fresh):
        scale = float(dw0.abs().max())
        np.testing.assert_allclose(dw.cpu().numpy(), dw0.cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)
        np.testing.assert_allclose(db.cpu().numpy(), db0.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(db0.abs().max()))


@pytest.mark.parametrize('case', [(2, 32, 32, (6, 128, 256), False), (2, 40, 72, (5, 126, 256), False), (8, 128, 128, (4, 16, 16), False),
                                  (8, 256, 64, (4, 16, 16), True)],
                         ids=['32to32', 'ragged', 'w16', 'w16ups'])
def test_wgrad_sliding_halo_kernel_on_the_16x16x32_mfma(case, sg_env, monkeypatch):
    """SG_WGRAD3L_16=1: conv_wgrad3l's K loop on v_mfma_f32_16x16x32_bf16 (K step = 32 voxels, four 16 x 16 accumulator tiles per tap).
    A diagnostic variant (measured slower inside the step than the 32x32x16 form: DESIGN_NOTES section 8) that must stay correct: the same
    sums up to f32 rounding (the tiles are added in another order)."""
    from saragan_amd import functional as F
    n, cin, cout, sp, ups = case
    dtype = torch.bfloat16
    xs = tuple(v // 2 for v in sp) if ups else sp
    x = cl(rnd((n, cin, *xs), 701, dtype), dtype)
    gy = cl(rnd((n, cout, *sp), 702, dtype), dtype)
    monkeypatch.setattr(F, '_NO_SUBPIXEL', True)
    dw0, db0 = F.raw_wgrad(x, gy, (3, 3, 3), 0.05, ups=ups, want_db=True)
    sg_env(SG_WGRAD3L_16=1)
    dw1, db1 = F.raw_wgrad(x, gy, (3, 3, 3), 0.05, ups=ups, want_db=True)
    np.testing.assert_allclose(dw1.cpu().numpy(), dw0.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(dw0.abs().max()))
    np.testing.assert_allclose(db1.cpu().numpy(), db0.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(db0.abs().max()))
    assert not torch.equal(dw0, torch.zeros_like(dw0))


@pytest.mark.parametrize('dtype', DT)
def test_to_rgb_filter_gradient_inside_the_pixel_norm_backward(dtype, monkeypatch):
    """sg_pixel_norm_act_bwd_pw_wg: to_rgb's filter and bias gradient (pgan/generator.py:13-16) from the pass that already reads the
    stage's output and the image gradient -- the same values, up to f32 summation order, as the separate filter-gradient pass over y
    (SARAGAN_NO_RGB_WG_FUSION=1), and the same stage gradients bit for bit."""
    from saragan_amd import functional as F
    n, cin, c, sp = 2, 16, 32, (4, 12, 16)
    x = cl(rnd((n, cin, *sp), 801, dtype), dtype).requires_grad_(True)
    w = rnd((3, 3, 3, cin, c), 802, dtype).float().to(dev()).requires_grad_(True)
    b = (rnd((c,), 803, torch.float32) * 0.3).float().to(dev()).requires_grad_(True)
    wr = rnd((1, 1, 1, c, 1), 804, dtype).float().to(dev()).requires_grad_(True)
    br = (rnd((1,), 805, torch.float32) * 0.3).float().to(dev()).requires_grad_(True)
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    coef_r = O.runtime_coef(wr.shape, 'linear', None)
    g_img = cl(rnd((n, 1, *sp), 806, dtype), dtype)

    def grads():
        img, _ = F.conv3d_pn_to_rgb(x, w, coef, b, False, 0.2, 1e-8, None, wr, coef_r, br)
        return torch.autograd.grad(img, [x, w, b, wr, br], g_img)

    import saragan_amd
    saragan_amd.set_deterministic(True)       # (the filter gradient's atomics would differ from run to run in the last bits)
    try:
        fused = grads()
        monkeypatch.setattr(F, '_NO_RGB_WG_FUSION', True)
        apart = grads()
    finally:
        saragan_amd.set_deterministic(False)
    for name, a, r in zip(('dx', 'dw', 'db'), fused[:3], apart[:3]):
        assert torch.equal(a, r), name
    for name, a, r in zip(('dw_rgb', 'db_rgb'), fused[3:], apart[3:]):
        np.testing.assert_allclose(a.float().cpu().numpy(), r.float().cpu().numpy(), rtol=2e-5, atol=2e-6 * float(r.abs().max()), err_msg=name)
        assert float(r.abs().max()) > 0
