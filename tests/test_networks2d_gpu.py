"""The 2-D pgan path (BASELINE config 5; SURFGAN_2D/networks/pgan/*.py with the legacy num_phases / base_dim / size
signature, SURFGAN_2D/networks/ops.py): images are D == 1 volumes on the same kernels.  The oracle is the 3-D
restatement on [N,C,1,H,W] with (1,3,3) kernels (oracle.specs_2d) and the 2-D tree's full-reduction gradient penalty."""
import numpy as np
import pytest
import torch

from oracle import pgan_oracle as O

pytestmark = pytest.mark.gpu
BASE = (3, 1, 4, 4)


def _to_oracle_names(params2d):
    """HWIO conv weights of the 2-D variables -> DHWIO with kD = 1."""
    return {k: (v.unsqueeze(0) if v.dim() == 4 else v) for k, v in params2d.items()}


def test_ops_2d_match_oracle():
    from saragan_amd import functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn((2, 16, 1, 12, 20), generator=g).to(dtype)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        up = F.upscale2x(xd, 1.0, factors=(1, 2, 2))
        ref = x.double().repeat_interleave(2, 3).repeat_interleave(2, 4)
        np.testing.assert_array_equal(up.detach().double().cpu().numpy(), ref.numpy())
        (gx,) = torch.autograd.grad(up, xd, torch.ones_like(up))
        assert float(gx.float().min()) == float(gx.float().max()) == 4.0       # gradient = 4 * avg_pool2d
        dn = F.downscale2x(xd, 0.25, None, factors=(1, 2, 2))
        refd = torch.nn.functional.avg_pool2d(x.double().squeeze(2), 2).unsqueeze(2)
        tol = 1e-6 if dtype == torch.float32 else 1e-2
        np.testing.assert_allclose(dn.detach().double().cpu().numpy(), refd.numpy(), rtol=tol, atol=tol)
        hp = F.downscale2x(xd, 0.5, None, factors=(1, 2, 1))                    # H-only pooling
        refh = 0.5 * (x.double()[:, :, :, 0::2] + x.double()[:, :, :, 1::2])
        np.testing.assert_allclose(hp.detach().double().cpu().numpy(), refh.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize('alpha', [0.0, 0.3])
def test_pgan_2d_step_matches_oracle(alpha):
    """One optimisation step of the 2-D pgan (4 phases -> 32x32 RGB, 'xxs' widths 16,8,4,2) through
    optimize_step with the spec-API adapters, fp32 HIP vs the fp64 oracle on D == 1 volumes."""
    from tests.cfgutil import assert_adam_close, build_product, pick, rel_l2
    from saragan_amd.networks2d.ops import num_filters
    from saragan_amd.networks2d.pgan.variables import legacy_spec, variable_shapes
    nph, size, phase, n, latent = 4, 'xxs', 4, 4, 32
    base_dim = num_filters(1, nph, size=size)
    spec = legacy_spec(nph, base_dim, size)
    ks, fs = O.specs_2d(nph, size)
    plan = variable_shapes(phase, BASE, latent, None, spec)
    want = O.variable_shapes(phase, BASE, latent, ks, fs)
    assert list(plan.keys()) == list(want.keys())
    for k_, shp in plan.items():
        assert tuple(want[k_]) == ((1, *shp) if len(shp) == 4 else tuple(shp)), k_
    p0 = O.init_params(phase, BASE, latent, ks, fs, seed=9, bias_std=0.05)
    img = (3, 1, 32, 32)
    rnd = O.draw_randomness(n, latent, img, 10)
    real = torch.randn((n, *img), dtype=torch.float64, generator=torch.Generator().manual_seed(11))
    cfg = dict(phase=phase, base_shape=BASE, latent_dim=latent, kernel_spec=ks, filter_spec=fs, activation='leaky_relu',
               leakiness=0.2, loss_fn='wgan', gp_weight=10.0, noise_stddev=0.01, two_d=True)
    freeze = list(O.variable_shapes(phase - 1, BASE, latent, ks, fs).keys()) if alpha > 0 else None
    p0_2d = {k_: (v.squeeze(0) if v.dim() == 5 else v) for k_, v in p0.items()}
    case = dict(p0=p0_2d, rnd=rnd, real=real, alpha=alpha, freeze=freeze, phase=phase, loss_fn='wgan', n=n, latent=latent,
                base=BASE, img=img, cfg=dict(cfg, kernel_spec=None, filter_spec=spec))
    store, tup, ph, ema, sess, _ = build_product(case, torch.float32, arch='networks2d.pgan.spec_api')
    tg, td, gg_h, gv, dg_h, dv, _, _ = pick(tup, freeze is not None)
    _, _, gl, dl, gs, gg, dg = sess.run([tg, td, tup[2], tup[3], tup[5], gg_h, dg_h], feed_dict={ph: real.float()})
    p = {k_: v.clone() for k_, v in p0.items()}
    ref = O.step_simultaneous(p, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None, rnd, real, alpha, cfg, 1e-3, 1e-3, freeze=freeze)
    np.testing.assert_allclose(float(gl), float(ref['gen_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(dl), float(ref['disc_loss']), rtol=1e-4, atol=1e-5)
    # (pixel-norm over 2-4 channels at the top levels of 'xxs' amplifies f32 summation error: 1e-3)
    np.testing.assert_allclose(gs.double().cpu().numpy(), ref['gen_sample'].numpy(), rtol=1e-3, atol=5e-4)     # (pixel-norm over 4 channels: one element in 6 x 10^6 was 2.4e-4 off)
    for hv, grads, refs in ((gv, gg, ref['g_grads']), (dv, dg, ref['d_grads'])):
        assert [v.key for v in hv] == list(refs.keys())
        for v, g_ in zip(hv, grads):
            assert rel_l2(g_, refs[v.key].reshape(g_.shape)) <= 1e-2, v.key      # 2-4 channel layers: f32 noise shows
    for k_, v in store.vars.items():
        assert_adam_close(v, p[k_].reshape(v.shape), 1e-3, 1e-4, k_, max_flip_frac=1e-3)


def test_pgan_2d_legacy_signature_1024():
    """BASELINE config 5 at its own size: the 1024^2 phase (9 phases from 4^2) of the 'xs' 2-D pgan in fp32 through
    the legacy signature, batch 2: shapes, finiteness, parameter count of the plan."""
    from saragan_amd.networks2d.ops import num_filters
    from saragan_amd.networks2d.pgan.discriminator import discriminator
    from saragan_amd.networks2d.pgan.generator import generator
    from saragan_amd.networks2d.pgan.variables import legacy_spec, variable_shapes
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    set_compute_dtype(torch.float32)
    nph, size, latent = 9, 'xs', 512
    base_dim = num_filters(1, nph, size=size)
    store = VariableStore('cuda', seed=3)
    with use_store(store), torch.no_grad():
        z = torch.randn(2, latent, device='cuda')
        img = generator(z, 0.5, 9, nph, base_dim, [3, 4, 4], 'leaky_relu', param=0.2, size=size)
        assert tuple(img.shape) == (2, 3, 1024, 1024) and torch.isfinite(img).all()
        out = discriminator(img, 0.5, 9, nph, base_dim, latent, 'leaky_relu', param=0.2, size=size)
        assert tuple(out.shape) == (2, 1) and torch.isfinite(out).all()
    plan = variable_shapes(9, (3, 1, 4, 4), latent, None, legacy_spec(nph, base_dim, size))
    assert {k_: tuple(v.shape) for k_, v in store.vars.items()} == {k_: tuple(v) for k_, v in plan.items()}
    assert plan['generator/generator_block_9/conv_2/weight'] == (3, 3, 4, 4)      # 4 filters at 1024^2 ('xs')


# ---------------------------------------------------------------------------------------------------
# BASELINE config 5 at its own size (1024 x 1024, 'xs', fp32): layers and the whole step
# ---------------------------------------------------------------------------------------------------
LAYERS_2D = [
    # tag, n, cin, cout, (h, w), kernel
    ('cfg5 D from_rgb 3->4 @1024^2', 2, 3, 4, (1024, 1024), (1, 1, 1)),
    ('cfg5 D conv_1 4->4 @1024^2', 2, 4, 4, (1024, 1024), (1, 3, 3)),
    ('cfg5 D conv_2 4->8 @1024^2', 2, 4, 8, (1024, 1024), (1, 3, 3)),
    ('cfg5 G conv_1 8->4 @1024^2', 2, 8, 4, (1024, 1024), (1, 3, 3)),
    ('cfg5 D conv_2 8->16 @512^2', 2, 8, 16, (512, 512), (1, 3, 3)),
    ('cfg5 D conv_2 16->32 @256^2', 2, 16, 32, (256, 256), (1, 3, 3)),
    ('cfg5 G conv_1 32->16 @256^2', 2, 32, 16, (256, 256), (1, 3, 3)),
]


@pytest.mark.parametrize('layer', LAYERS_2D, ids=[l[0] for l in LAYERS_2D])
def test_config5_layers_exact_shapes(layer):
    """conv2d + bias + LeakyReLU forward, data gradient and weight gradient of the small-channel layers of the 1024^2
    phase (SURFGAN_2D/networks/ops.py:99-102) at their exact shapes, fp32, against the oracle: forward on boxes of the
    image (corners, faces, interior, tile seams), backward with a DENSE upstream gradient over the whole image against
    torch's CPU convolution backward in fp64 (these layers are small enough for the full-tensor oracle)."""
    import torch.nn.functional as TF
    from saragan_amd import functional as F
    from tests.test_configs_gpu import _boxes, _oracle_box
    tag, n, cin, cout, (h, w_), k = layer
    sp = (1, h, w_)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(len(tag) * 7 + cin)
    x = torch.randn((n, cin, *sp), generator=g)
    w = torch.randn((*k, cin, cout), generator=g)
    b = torch.randn(cout, generator=g) * 0.1
    coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
    wt = (w.double() * coef).permute(4, 3, 0, 1, 2).contiguous()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = F.conv3d(xd, wd, coef, bias=bd, act=True, slope=0.2)
    assert tuple(y.shape) == (n, cout, *sp)
    yc = y.detach().double().cpu()
    scale = float(yc.abs().max())
    for lo, hi in _boxes(sp, k):
        ref = _oracle_box(x.double(), wt, lo, hi, k) + b.double().reshape(1, -1, 1, 1, 1)
        ref = torch.maximum(ref, ref * 0.2)
        got = yc[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-5 * scale, err_msg=f'{tag} fwd box {lo}')
    # whole-image backward, dense upstream gradient
    xr = x.double().requires_grad_(True)
    wr = (w.double() * coef).requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = O.leaky_relu(TF.conv3d(xr, wr.permute(4, 3, 0, 1, 2), padding=(0, k[1] // 2, k[2] // 2)) + br.reshape(1, -1, 1, 1, 1), 0.2)
    np.testing.assert_allclose(yc.numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-5 * scale, err_msg=f'{tag} fwd')
    gy = torch.randn((n, cout, *sp), generator=g)
    gy = torch.where(yr.detach().abs() < 1e-4 * scale, torch.zeros_like(gy), gy.double()).float()   # (mask flips: see test_configs_gpu)
    gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], gy.to(dev).contiguous(memory_format=torch.channels_last_3d))
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], gy.double())
    from tests.cfgutil import rel_l2
    np.testing.assert_allclose(gx.detach().double().cpu().numpy(), gxr.numpy(), rtol=1e-4, atol=1e-5 * float(gxr.abs().max()),
                               err_msg=f'{tag} dgrad')
    refw = (gwr * coef).numpy()
    # 2 x 10^6 terms per weight-gradient element summed in f32: 1e-4 of the largest element
    np.testing.assert_allclose(gw.detach().double().cpu().numpy(), refw, rtol=1e-4, atol=1e-4 * np.abs(refw).max(), err_msg=f'{tag} wgrad')
    np.testing.assert_allclose(gb.detach().double().cpu().numpy(), gbr.numpy(), rtol=1e-4, atol=1e-4 * float(gbr.abs().max()),
                               err_msg=f'{tag} bias grad')
    assert rel_l2(gw, refw) <= 1e-4 and rel_l2(gx, gxr) <= 1e-5


def test_config5_full_size_step_matches_oracle():
    """BASELINE configs[4] at its own size: ONE WHOLE G+D step of the 'xs' 2-D pgan at 1024 x 1024 RGB (phase 9 of 9,
    latent 512, wgan-gp, alpha 0), batch 2, fp32 HIP against the fp64 oracle at the same size (~40 s of CPU): losses,
    sample, every gradient and the post-Adam weights; then the step again from the same start -- the weight-gradient
    sums are atomic, so gradient norms repeat to 1e-4, not bit for bit; previous-phase to_rgb / from_rgb (faded out at
    alpha 0) get exactly zero gradients and every other variable moves."""
    from tests.cfgutil import assert_adam_close, build_product, pick, rel_l2
    from saragan_amd.networks2d.ops import num_filters
    from saragan_amd.networks2d.pgan.variables import legacy_spec
    nph, size, phase, n, latent = 9, 'xs', 9, 2, 512
    spec = legacy_spec(nph, num_filters(1, nph, size=size), size)
    ks, fs = O.specs_2d(nph, size)
    p0 = O.init_params(phase, BASE, latent, ks, fs, seed=19, bias_std=0.05)
    img = (3, 1, 1024, 1024)
    rnd = O.draw_randomness(n, latent, img, 20)
    rng = np.random.default_rng(21)
    real = torch.as_tensor((np.clip(rng.normal(1024, 512, (n, *img)), 0, 4095).astype(np.int16).astype(np.float64) - 1024.0) / 1024.0)
    cfg = dict(phase=phase, base_shape=BASE, latent_dim=latent, kernel_spec=ks, filter_spec=fs, activation='leaky_relu',
               leakiness=0.2, loss_fn='wgan', gp_weight=10.0, noise_stddev=0.01, two_d=True)
    p0_2d = {k_: (v.squeeze(0) if v.dim() == 5 else v) for k_, v in p0.items()}
    case = dict(p0=p0_2d, rnd=rnd, real=real, alpha=0.0, freeze=None, phase=phase, loss_fn='wgan', n=n, latent=latent,
                base=BASE, img=img, cfg=dict(cfg, kernel_spec=None, filter_spec=spec))
    torch.set_num_threads(min(16, torch.get_num_threads() * 2))
    p = {k_: v.clone() for k_, v in p0.items()}
    ref = O.step_simultaneous(p, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None, rnd, real, 0.0, cfg, 1e-3, 1e-3)
    runs = []
    for rep in range(2):
        store, tup, ph, ema, sess, _ = build_product(case, torch.float32, arch='networks2d.pgan.spec_api')
        tg, td, gg_h, gv, dg_h, dv, _, _ = pick(tup, False)
        _, _, gl, dl, gs, gg, dg = sess.run([tg, td, tup[2], tup[3], tup[5], gg_h, dg_h], feed_dict={ph: real.float()})
        assert tuple(gs.shape) == (n, 3, 1, 1024, 1024) or tuple(gs.shape) == (n, 3, 1024, 1024)
        runs.append({v.key: g_.detach().double().cpu() for hv, grads in ((gv, gg), (dv, dg)) for v, g_ in zip(hv, grads)})
        if rep == 0:
            np.testing.assert_allclose(float(gl), float(ref['gen_loss']), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(float(dl), float(ref['disc_loss']), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(gs.double().cpu().reshape(ref['gen_sample'].shape).numpy(), ref['gen_sample'].numpy(),
                                       rtol=1e-3, atol=5e-4)     # (pixel-norm over 4 channels: one element in 6 x 10^6 was 2.4e-4 off)
            report = {}
            for hv, grads, refs in ((gv, gg, ref['g_grads']), (dv, dg, ref['d_grads'])):
                assert [v.key for v in hv] == list(refs.keys())
                for v, g_ in zip(hv, grads):
                    r = refs[v.key].reshape(g_.shape)
                    if float(r.abs().max()) == 0.0:
                        assert float(g_.abs().max()) == 0.0, v.key
                        assert 'rgb_8' in v.key, v.key          # only the faded-out branch has zero gradients
                        continue
                    report[v.key] = rel_l2(g_, r)
            print('config 5 full-size gradients, rel L2 vs fp64 oracle:', {k_: round(v, 6) for k_, v in report.items()})
            assert max(report.values()) <= 1e-2, report
            for k_, v in store.vars.items():
                assert_adam_close(v, p[k_].reshape(v.shape), 1e-3, 1e-4, k_, max_flip_frac=5e-3)   # (as config 2: near-zero gradients)
                moved = not torch.equal(v.detach().cpu().double().reshape(-1), p0[k_].reshape(-1).float().double())
                assert moved == ('rgb_8' not in k_), k_
        del store, tup, sess, ema
    for k_, a in runs[0].items():
        assert rel_l2(runs[1][k_], a) <= 1e-4 if float(a.abs().max()) > 0 else float(runs[1][k_].abs().max()) == 0.0, k_
    from saragan_amd import functional as F
    F.clear_pack_cache()
    torch.cuda.empty_cache()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cin,cout', [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (4, 16), (16, 4), (16, 16), (16, 32)])
def test_small_channel_kernels_match_the_mfma_path(cin, cout, dtype, sg_env):
    """The small-channel VALU kernels (csrc/small.hip: conv_small_fwd / conv_small_wgrad) against the same calls through the
    MFMA kernels (SG_NO_SMALL=1) on a ragged 2-D shape (W not a multiple of the 128-pixel segment): forward with bias +
    LeakyReLU + pixel-norm + sign words + scale, data gradient with a LeakyReLU mask in the epilogue, weight and bias
    gradient -- and that the small kernels are the ones that ran."""
    import ctypes as C
    from saragan_amd import _lib
    from saragan_amd import functional as F
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(cin * 17 + cout)
    n, h, w_ = 3, 150, 202
    x = torch.randn((n, cin, 1, h, w_), generator=g).to(dtype).to(dev).contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn((n, cout, 1, h, w_), generator=g).to(dtype).to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((1, 3, 3, cin, cout), generator=g).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    coef = float(O.runtime_coef(w.shape, 'leaky_relu', 0.2))

    def run():
        lib.sg_prof_enable(1)
        y, scale, signs = F.raw_conv(x, w, coef, False, bias=b, act=True, slope=0.2, pixel_norm=True, want_scale=True, want_signs=True)
        y2, _, signs2 = F.raw_conv(x, w, coef, False, bias=b, act=True, slope=0.2, want_signs=True)
        gx, _, _ = F.raw_conv(gy, w, coef, True, mask_bits=F.sign_words(x), mask_slope=0.2)
        dw, db = F.raw_wgrad(x, gy, (1, 3, 3), coef, want_db=True)
        torch.cuda.synchronize()
        ents = (_lib.ProfEntry * 64)()
        cnt = C.c_int32(0)
        lib.sg_prof_collect(ents, 64, C.byref(cnt))
        lib.sg_prof_enable(0)
        F.clear_pack_cache()
        return (y, scale, signs, y2, signs2, gx, dw, db), sorted({ents[i].kernel.decode() for i in range(cnt.value)})

    got, kern = run()
    # which layers the small kernels take: forward / data gradient up to cin * cout = 128; weight gradient those with cin <= 8,
    # and 16 input channels (as two groups of 8) with 8 or 16 outputs
    small_fwd = cin * cout <= 128
    small_wg = (small_fwd and cin <= 8) or (cin == 16 and cout in (8, 16))
    assert any('conv_small_fwd' in k for k in kern) == small_fwd and any('conv_small_wgrad' in k for k in kern) == small_wg, kern
    sg_env(SG_NO_SMALL=1)
    ref, kern_ref = run()
    assert not any('conv_small' in k for k in kern_ref), kern_ref
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    names = ('y (pixel-norm)', 'pn scale', 'sign words', 'y', 'sign words (no pn)', 'masked data gradient', 'dw', 'db')
    for name, a_, r_ in zip(names, got, ref):
        if a_.dtype == torch.int32:
            # a sign may differ where the pre-activation is within rounding of zero
            assert float((a_ != r_).float().mean()) <= (1e-4 if dtype == torch.float32 else 5e-3), name
            continue
        a64, r64 = a_.double(), r_.double()
        err = float((a64 - r64).abs().max() / r64.abs().max())
        assert err <= (tol * (50 if name in ('dw', 'db') and dtype == torch.float32 else 1)), (name, err)
    # and the weight gradient against torch's fp64 convolution backward (both paths sum ~10^5 terms in f32)
    xr = x.double().cpu().squeeze(2)
    wr = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    (gw,) = torch.autograd.grad(torch.nn.functional.conv2d(xr, wr, padding=1), wr, gy.double().cpu().squeeze(2))
    refw = gw.permute(2, 3, 1, 0).unsqueeze(0) * coef
    assert float((got[6].double().cpu() - refw).abs().max() / refw.abs().max()) <= (1e-4 if dtype == torch.float32 else 1e-2)
    # run-to-run: the slab reduction has no atomics
    again, _ = (sg_env(SG_NO_SMALL=0), run())[1]
    if small_wg:
        assert torch.equal(again[6], got[6]) and torch.equal(again[7], got[7])
