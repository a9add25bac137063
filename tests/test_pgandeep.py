"""pgandeep (reference networks/pgandeep/*.py): N = len(kernel_spec[phase]) convolutions per
block behind the same generator / discriminator signature, so the same optimize_step drives it."""
import numpy as np
import pytest
import torch

from oracle import pgan_oracle as O

BASE = (1, 1, 4, 4)
KS = [[[1, 3, 3]] * 3, [[1, 3, 3], [3, 3, 3], [3, 3, 3]], [[3, 3, 3]] * 3]
FS = [[16, 16, 16], [16, 8, 8], [8, 8, 8]]
LATENT = 16


def test_variable_plan_matches_oracle_and_reference_quirks():
    from saragan_amd.networks.pgandeep.variables import variable_shapes
    for phase in (1, 2, 3):
        want = O.variable_shapes_deep(phase, BASE, LATENT, KS, FS)
        got = variable_shapes(phase, BASE, LATENT, KS, FS)
        assert list(want.items()) == [(k, tuple(v)) for k, v in got.items()]
    plan = variable_shapes(3, BASE, LATENT, KS, FS)
    assert 'generator/generator_in/conv_2/weight' in plan and 'generator/generator_in/conv/weight' not in plan
    assert plan['discriminator/discriminator_block_3/conv_1/weight'][:3] == (3, 3, 3)      # kernel_spec[2][1] for all
    assert plan['discriminator/discriminator_block_2/conv_3/weight'][-1] == FS[0][2]       # last layer: previous phase
    assert plan['discriminator/discriminator_out/conv_2/weight'][:3] == tuple(KS[0][1])
    with pytest.raises(ValueError):
        variable_shapes(4, BASE, LATENT, KS, FS)
    with pytest.raises(ValueError):      # a spec whose fade-in branches disagree in width
        variable_shapes(3, BASE, LATENT, KS, [[16, 16, 16], [16, 16, 8], [8, 8, 8]])


def test_two_layer_pgandeep_generator_is_pgan_up_to_names():
    ks2 = [[[1, 3, 3], [1, 3, 3]], [[1, 3, 3], [3, 3, 3]], [[3, 3, 3], [3, 3, 3]]]
    fs2 = [[16, 16], [16, 8], [8, 8]]
    pd = O.init_params(3, BASE, LATENT, ks2, fs2, arch='pgandeep', seed=3)
    pp = {k.replace('generator_in/conv_1', 'generator_in/conv'): v for k, v in pd.items()}
    z = torch.randn(2, LATENT, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    a = O.generator_deep(pd, z, 0.3, 3, BASE, 'leaky_relu', ks2, fs2, 0.2)
    b = O.generator(pp, z, 0.3, 3, BASE, 'leaky_relu', ks2, fs2, 0.2)
    assert float((a - b).abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('alpha,loss_fn', [(0.0, 'wgan'), (0.4, 'logistic')])
def test_pgandeep_step_matches_oracle(alpha, loss_fn):
    """One optimisation step of the three-convolutions-per-block network, fp32 HIP vs the fp64 oracle: losses, sample,
    every gradient, post-Adam weights (stabilising and mixing / freeze)."""
    from tests.cfgutil import assert_adam_close, build_product, pick, rel_l2
    phase, n = 3, 4
    img = (1, 4, 16, 16)
    p0 = O.init_params(phase, BASE, LATENT, KS, FS, seed=5, bias_std=0.05, arch='pgandeep')
    rnd = O.draw_randomness(n, LATENT, img, 6)
    real = torch.randn((n, *img), dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    cfg = dict(phase=phase, base_shape=BASE, latent_dim=LATENT, kernel_spec=KS, filter_spec=FS, activation='leaky_relu',
               leakiness=0.2, loss_fn=loss_fn, gp_weight=10.0 if loss_fn == 'wgan' else 1.0, noise_stddev=0.01,
               arch='pgandeep')
    freeze = list(O.variable_shapes_deep(phase - 1, BASE, LATENT, KS, FS).keys()) if alpha > 0 else None
    case = dict(p0=p0, rnd=rnd, real=real, alpha=alpha, cfg=cfg, freeze=freeze, phase=phase, loss_fn=loss_fn, n=n,
                latent=LATENT, base=BASE, img=img)
    store, tup, ph, ema, sess, _ = build_product(case, torch.float32, arch='pgandeep')
    assert list(store.vars.keys()) == list(p0.keys())
    tg, td, gg_h, gv, dg_h, dv, _, _ = pick(tup, freeze is not None)
    _, _, gl, dl, gs, gg, dg = sess.run([tg, td, tup[2], tup[3], tup[5], gg_h, dg_h], feed_dict={ph: real.float()})
    p = {k: v.clone() for k, v in p0.items()}
    ref = O.step_simultaneous(p, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None, rnd, real, alpha, cfg, 1e-3, 1e-3,
                              freeze=freeze)
    np.testing.assert_allclose(float(gl), float(ref['gen_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(dl), float(ref['disc_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gs.double().cpu().numpy(), ref['gen_sample'].numpy(), rtol=1e-4, atol=1e-5)
    for hv, grads, refs in ((gv, gg, ref['g_grads']), (dv, dg, ref['d_grads'])):
        assert [v.key for v in hv] == list(refs.keys())
        for v, g in zip(hv, grads):
            assert rel_l2(g, refs[v.key]) <= 1e-3, v.key
    for k, v in store.vars.items():
        assert_adam_close(v, p[k], 1e-3, 1e-4, k)
