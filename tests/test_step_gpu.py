"""GPU parity of the whole hot path (networks -> losses -> gradients -> TF-Adam -> EMA) against the fp64
oracle fixtures tests/golden/oracle_step_*.npz, driven through the reference-shaped API
(optimization.optimize_step + Session.run).  fp32 path tolerances follow SURVEY.md section 8c:
rtol 1e-4 / atol 1e-5 on activations and losses, 1e-3 on gradients (they pass through the GP double backward)."""
import os

import numpy as np
import pytest
import torch

from tests.stepfix import BASE_SHAPE, FILTER_SPEC, KERNEL_SPEC, LATENT, load_step_fixture

pytestmark = pytest.mark.gpu

FIXTURES = ['oracle_step_p1_wgan_a000.npz', 'oracle_step_p2_wgan_a060.npz', 'oracle_step_p3_logistic_a025.npz',
            'oracle_step_p3_wgan_a000.npz']


def _build(fx, dtype, strategy='simultaneous'):
    import saragan_amd.optimization as opt
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    set_compute_dtype(dtype)
    store = VariableStore('cuda', seed=0)
    L.set_random_source(L.InjectedRandom({k: v.float() for k, v in fx['rnd'].items()}))
    alpha = ScalarVariable(fx['alpha'], 'alpha')
    g_lr, d_lr = ScalarVariable(1e-3, 'g_lr'), ScalarVariable(1e-3, 'd_lr')
    og, od = opt.AdamOptimizer(g_lr, 0.0, 0.9), opt.AdamOptimizer(d_lr, 0.0, 0.9)
    ph = opt.Placeholder([4, 1, 1, 1, 1])
    cfg = fx['cfg']
    freeze = None if fx['freeze'] is None else list(fx['freeze'])
    with use_store(store):
        tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, alpha, fx['phase'], BASE_SHAPE,
                                KERNEL_SPEC, FILTER_SPEC, 'leaky_relu', 0.2, fx['loss_fn'], cfg['gp_weight'],
                                strategy, False, False, 0.01, freeze if freeze is not None else None)
    store.load_state_dict({k: v for k, v in fx['p0'].items()}, strict=True)
    graph = tup[0].graph
    ema = ExtendedEMA(list(store.vars.keys()), 0.99, graph=graph)
    return store, tup, ph, ema, opt.Session('cuda')


@pytest.mark.parametrize('name', FIXTURES)
def test_step_matches_oracle_fp32(golden_dir, name):
    fx = load_step_fixture(os.path.join(golden_dir, name), torch.float64)
    store, tup, ph, ema, sess = _build(fx, torch.float32)
    (train_gen, train_disc, gen_loss, disc_loss, gp_loss, gen_sample, g_grad, g_vars, d_grad, d_vars, mg, md,
     train_gen_fz, g_grad_fz, g_vars_fz, mg_fz, train_disc_fz, d_grad_fz, d_vars_fz, md_fz) = tup
    mixing = fx['freeze'] is not None
    tg, td = (train_gen_fz, train_disc_fz) if mixing else (train_gen, train_disc)
    gg_h, gv, dg_h, dv = (g_grad_fz, g_vars_fz, d_grad_fz, d_vars_fz) if mixing else (g_grad, g_vars, d_grad, d_vars)
    ema_op = ema.apply()
    feed = {ph: fx['real'].float()}
    _, _, gl, dl, gpl, gs, gg, dg, mgn, mdn = sess.run(
        [tg, td, gen_loss, disc_loss, gp_loss, gen_sample, gg_h, dg_h, (mg_fz if mixing else mg),
         (md_fz if mixing else md)], feed_dict=feed)
    sess.run(ema_op)
    np.testing.assert_allclose(float(gl), float(fx['gen_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(dl), float(fx['disc_loss']), rtol=1e-4, atol=1e-5)
    ref = fx['gp_loss'].numpy()
    np.testing.assert_allclose(gpl.double().cpu().numpy().reshape(ref.shape), ref, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(gs.double().cpu().numpy(), fx['gen_sample'].numpy(), rtol=1e-4, atol=1e-5)
    for handle_vars, grads, refs in ((gv, gg, fx['gg']), (dv, dg, fx['dg'])):
        assert [v.key for v in handle_vars] == list(refs.keys())
        for v, g in zip(handle_vars, grads):
            r = refs[v.key].numpy()
            np.testing.assert_allclose(g.double().cpu().numpy(), r, rtol=1e-3, atol=1e-4 * np.abs(r).max() + 1e-9,
                                       err_msg=v.key)
    want_max = max(float(torch.linalg.vector_norm(v)) for v in fx['dg'].values())
    np.testing.assert_allclose(float(mdn), want_max, rtol=1e-3)
    # step 1 weights + EMA, then a second step
    def check_params(tag_p, tag_e):
        for k, p in store.vars.items():
            r = fx[tag_p][k].numpy()
            np.testing.assert_allclose(p.detach().double().cpu().numpy(), r, rtol=1e-4, atol=2e-5, err_msg=f'{tag_p}:{k}')
            e = fx[tag_e][k].numpy()
            np.testing.assert_allclose(ema.average(k).double().cpu().numpy(), e, rtol=1e-4, atol=2e-5, err_msg=f'{tag_e}:{k}')
    check_params('p1', 'ema1')
    sess.run([tg, td], feed_dict=feed)
    sess.run(ema_op)
    check_params('p2', 'ema2')
    if mixing:   # quirk Q4: previous-phase variables are not updated while alpha > 0
        for k in fx['freeze']:
            if k in store.vars:
                np.testing.assert_array_equal(store.vars[k].detach().cpu().numpy(), fx['p0'][k].float().numpy())


@pytest.mark.parametrize('name', FIXTURES[1:3])
def test_step_bf16_close_to_oracle(golden_dir, name):
    """bf16 storage/MFMA path vs the fp64 oracle: rtol 2e-2 on losses and samples."""
    fx = load_step_fixture(os.path.join(golden_dir, name), torch.float64)
    store, tup, ph, ema, sess = _build(fx, torch.bfloat16)
    mixing = fx['freeze'] is not None
    tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
    _, _, gl, dl, gs = sess.run([tg, td, tup[2], tup[3], tup[5]], feed_dict={ph: fx['real'].float()})
    np.testing.assert_allclose(float(gl), float(fx['gen_loss']), rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(float(dl), float(fx['disc_loss']), rtol=3e-2, atol=3e-2)
    ref = fx['gen_sample'].numpy()
    np.testing.assert_allclose(gs.double().cpu().numpy(), ref, rtol=3e-2, atol=3e-2 * np.abs(ref).max())
    from saragan_amd.varstore import set_compute_dtype
    set_compute_dtype(torch.float32)


@pytest.mark.parametrize('name', [FIXTURES[1], FIXTURES[2]])   # wgan while mixing (freeze ops), logistic
def test_alternate_step_matches_oracle_fp32(golden_dir, name):
    """optim_strategy 'alternate' (optimization.py:165-220): D step from forward_discriminator's loss (GP over all of
    (c,d,h,w), loss.py:79), then the G loss on the UPDATED discriminator.  The oracle is replayed here in fp64 from
    the fixture's initial weights and randomness."""
    from oracle import pgan_oracle as O
    fx = load_step_fixture(os.path.join(golden_dir, name), torch.float64)
    store, tup, ph, ema, sess = _build(fx, torch.float32, strategy='alternate')
    mixing = fx['freeze'] is not None
    tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
    gg_h, dg_h = (tup[13], tup[17]) if mixing else (tup[6], tup[8])
    gv, dv = (tup[14], tup[18]) if mixing else (tup[7], tup[9])
    _, _, gl, dl, gs, gg, dg = sess.run([tg, td, tup[2], tup[3], tup[5], gg_h, dg_h], feed_dict={ph: fx['real'].float()})
    p = {k: v.clone() for k, v in fx['p0'].items()}
    ref = O.step_alternate(p, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None, fx['rnd'], fx['real'], fx['alpha'],
                           fx['cfg'], 1e-3, 1e-3, freeze=fx['freeze'])
    np.testing.assert_allclose(float(dl), float(ref['disc_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(gl), float(ref['gen_loss']), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gs.double().cpu().numpy(), ref['gen_sample'].numpy(), rtol=1e-4, atol=1e-5)
    for handle_vars, grads, refs in ((gv, gg, ref['g_grads']), (dv, dg, ref['d_grads'])):
        assert [v.key for v in handle_vars] == list(refs.keys())
        for v, g in zip(handle_vars, grads):
            r = refs[v.key].numpy()
            np.testing.assert_allclose(g.double().cpu().numpy(), r, rtol=1e-3, atol=1e-4 * np.abs(r).max() + 1e-9,
                                       err_msg=v.key)
    for k, v in store.vars.items():   # weights after the D-then-G updates
        np.testing.assert_allclose(v.detach().double().cpu().numpy(), p[k].numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


def test_variable_names_created_by_networks_match_plan():
    """Running generator()/discriminator() creates exactly the planned variables."""
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.networks.pgan.variables import pgan_variable_shapes
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    set_compute_dtype(torch.float32)
    store = VariableStore('cuda', seed=1)
    with use_store(store), torch.no_grad():
        z = torch.randn(2, LATENT, device='cuda')
        img = generator(z, 0.5, 3, BASE_SHAPE, 'leaky_relu', KERNEL_SPEC, FILTER_SPEC, param=0.2)
        assert tuple(img.shape) == (2, 1, 4, 16, 16)
        out = discriminator(img, 0.5, 3, LATENT, 'leaky_relu', KERNEL_SPEC, FILTER_SPEC, param=0.2)
        assert tuple(out.shape) == (2, 1)
        with pytest.raises(NotImplementedError):
            generator(z, 0.5, 3, BASE_SHAPE, 'leaky_relu', KERNEL_SPEC, FILTER_SPEC, param=0.2, conditioning=1)
        with pytest.raises(ValueError):
            generator(z, 0.5, 4, BASE_SHAPE, 'leaky_relu', KERNEL_SPEC, FILTER_SPEC, param=0.2)
    plan = pgan_variable_shapes(3, BASE_SHAPE, LATENT, KERNEL_SPEC, FILTER_SPEC)
    created = {k: tuple(v.shape) for k, v in store.vars.items()}
    created.pop('generator/generator_block_4/conv_1/weight', None)   # partial creation by the failing call
    assert {k: tuple(v) for k, v in plan.items()} == {k: v for k, v in created.items() if k in plan}
    assert set(created) - set(plan) <= {'generator/to_rgb_3/weight', 'generator/to_rgb_3/bias'}


@pytest.mark.parametrize('name', FIXTURES[1:])
def test_step_bf16_every_gradient_weight_and_ema(golden_dir, name):
    """bf16 storage / MFMA (the mode the bench number is quoted in) against the fp64 fixtures: EVERY G and D gradient
    (relative L2 per tensor; they pass through the GP double backward), the post-Adam weights and the EMA shadows after
    one and two steps.  With beta1 = 0 a first update is -lr * sign(g): elements whose gradient is below bf16 noise may
    land 2 lr away, at most 5 % of a tensor."""
    from tests.cfgutil import bf16_gradient_report
    fx = load_step_fixture(os.path.join(golden_dir, name), torch.float64)
    store, tup, ph, ema, sess = _build(fx, torch.bfloat16)
    mixing = fx['freeze'] is not None
    tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
    gg_h, gv, dg_h, dv = (tup[13], tup[14], tup[17], tup[18]) if mixing else (tup[6], tup[7], tup[8], tup[9])
    feed = {ph: fx['real'].float()}
    _, _, gg, dg = sess.run([tg, td, gg_h, dg_h], feed_dict=feed)
    sess.run(ema.apply())
    report, bad = bf16_gradient_report([('G', gv, gg, fx['gg']), ('D', dv, dg, fx['dg'])])
    print(name, {k: v for k, v in report.items() if k.endswith(':all')})
    assert not bad, (bad, report)
    flips = {}
    for k, p in store.vars.items():
        d = (p.detach().double().cpu() - fx['p1'][k]).abs()
        assert float(d.max()) <= 2.2e-3 + 1e-3 * float(fx['p1'][k].abs().max()), (k, float(d.max()))
        flips[k] = float((d > 1e-3).double().mean())
        if k.endswith('weight'):
            assert flips[k] <= 0.10, (k, flips[k])       # share of elements whose first Adam step went the other way
    sess.run([tg, td], feed_dict=feed)
    sess.run(ema.apply())
    for k, p in store.vars.items():   # second step: both arithmetic's weights have moved, differences add up
        d = (p.detach().double().cpu() - fx['p2'][k]).abs().max()
        # (a second Adam step moves an element by up to 1.38 lr: 2 lr + 2 * 1.38 lr when both steps went opposite ways)
        assert float(d) <= 5e-3 + 1e-3 * float(fx['p2'][k].abs().max()), (k, float(d))
    from saragan_amd.varstore import set_compute_dtype
    set_compute_dtype(torch.float32)


@pytest.mark.parametrize('name', FIXTURES[1:])
def test_step_bf16_against_the_bf16_emulating_oracle(golden_dir, name):
    """The same bf16 step against the oracle run with the HIP path's rounding points (oracle.bf16_emulation): masks come
    from the same values on both sides, so the per-tensor bound is 0.05 (weights) instead of the 0.30 the fp64
    comparison needs, losses within 2e-3 and the sample within 1e-2 relative L2."""
    from tests.cfgutil import bf16_emulated_step, bf16_emulation_report, rel_l2
    fx = load_step_fixture(os.path.join(golden_dir, name), torch.float64)
    ref = bf16_emulated_step(fx['p0'], fx['rnd'], fx['real'], fx['alpha'], fx['cfg'], fx['freeze'])
    store, tup, ph, ema, sess = _build(fx, torch.bfloat16)
    mixing = fx['freeze'] is not None
    tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
    gg_h, gv, dg_h, dv = (tup[13], tup[14], tup[17], tup[18]) if mixing else (tup[6], tup[7], tup[8], tup[9])
    _, _, gl, dl, gs, gg, dg = sess.run([tg, td, tup[2], tup[3], tup[5], gg_h, dg_h], feed_dict={ph: fx['real'].float()})
    from saragan_amd.varstore import set_compute_dtype
    set_compute_dtype(torch.float32)
    report, bad = bf16_emulation_report([('G', gv, gg, ref['g_grads']), ('D', dv, dg, ref['d_grads'])])
    print(name, 'gen_loss', float(gl), float(ref['gen_loss']), 'disc_loss', float(dl), float(ref['disc_loss']),
          'sample', rel_l2(gs, ref['gen_sample']), report)
    np.testing.assert_allclose(float(gl), float(ref['gen_loss']), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(float(dl), float(ref['disc_loss']), rtol=2e-3, atol=2e-3)
    assert rel_l2(gs, ref['gen_sample']) <= 1e-2
    assert not bad, (bad, report)


@pytest.mark.parametrize('name', [FIXTURES[3], FIXTURES[1]])
def test_global_norm_clipping_matches_oracle(golden_dir, name):
    """--g_clipping / --d_clipping (optimization.py:66-71): tf.clip_by_global_norm(grads, 1.0) before the update, and
    the max per-variable norm of the CLIPPED gradients; oracle.clip_by_global_norm replayed in fp64."""
    import saragan_amd.optimization as opt
    from oracle import pgan_oracle as O
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    fx = load_step_fixture(os.path.join(golden_dir, name), torch.float64)
    # scale the initial weights up so that both global norms exceed 1 and the clip is active
    p0 = {k: v * (1.5 if k.endswith('weight') else 1.0) for k, v in fx['p0'].items()}
    set_compute_dtype(torch.float32)
    store = VariableStore('cuda', seed=0)
    L.set_random_source(L.InjectedRandom({k: v.float() for k, v in fx['rnd'].items()}))
    alpha = ScalarVariable(fx['alpha'], 'alpha')
    og = opt.AdamOptimizer(ScalarVariable(1e-3, 'g_lr'), 0.0, 0.9)
    od = opt.AdamOptimizer(ScalarVariable(1e-3, 'd_lr'), 0.0, 0.9)
    ph = opt.Placeholder([4, 1, 1, 1, 1])
    cfg = fx['cfg']
    with use_store(store):
        tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, alpha, fx['phase'], BASE_SHAPE, KERNEL_SPEC,
                                FILTER_SPEC, 'leaky_relu', 0.2, fx['loss_fn'], cfg['gp_weight'], 'simultaneous', True, True,
                                0.01, None if fx['freeze'] is None else list(fx['freeze']))
    store.load_state_dict(p0, strict=True)
    mixing = fx['freeze'] is not None
    tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
    mg_h, md_h = (tup[15], tup[19]) if mixing else (tup[10], tup[11])
    sess = opt.Session('cuda')
    dg_h, dv = (tup[17], tup[18]) if mixing else (tup[8], tup[9])
    _, _, mgn, mdn, dg = sess.run([tg, td, mg_h, md_h, dg_h], feed_dict={ph: fx['real'].float()})
    p = {k: v.clone() for k, v in p0.items()}
    # unclipped gradients first, to make sure the clip bites in this case
    probe = O.step_simultaneous({k: v.clone() for k, v in p0.items()}, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None, fx['rnd'],
                                fx['real'], fx['alpha'], fx['cfg'], 1e-3, 1e-3, freeze=fx['freeze'])
    gn_g = float(torch.sqrt(sum((g * g).sum() for g in probe['g_grads'].values())))
    gn_d = float(torch.sqrt(sum((g * g).sum() for g in probe['d_grads'].values())))
    assert gn_d > 1.0, gn_d
    ref = O.step_simultaneous(p, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None, fx['rnd'], fx['real'], fx['alpha'], fx['cfg'],
                              1e-3, 1e-3, freeze=fx['freeze'], g_clipping=True, d_clipping=True)
    want_g = max(float(torch.linalg.vector_norm(g)) for g in ref['g_grads'].values())
    want_d = max(float(torch.linalg.vector_norm(g)) for g in ref['d_grads'].values())
    np.testing.assert_allclose(float(mgn), want_g, rtol=1e-3)
    np.testing.assert_allclose(float(mdn), want_d, rtol=1e-3)
    assert want_d <= 1.0 + 1e-9 and (gn_g <= 1.0 or want_g <= 1.0 + 1e-9)
    for v, g in zip(dv, dg):      # the returned gradients are the CLIPPED ones (optimization.py:66-75)
        r = ref['d_grads'][v.key].numpy()
        np.testing.assert_allclose(g.double().cpu().numpy(), r, rtol=2e-3, atol=1e-4 * np.abs(r).max() + 1e-9, err_msg=v.key)
    for k, v in store.vars.items():
        np.testing.assert_allclose(v.detach().double().cpu().numpy(), p[k].numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize('kind', ['SGD', 'Momentum', 'Adadelta'])
@pytest.mark.parametrize('strategy', ['simultaneous', 'alternate'])
def test_other_optimizers_match_oracle(golden_dir, kind, strategy):
    """--optimizer SGD / Momentum (Nesterov) / Adadelta (optimization.py:17-22,29-35) on the fused sg_optim_step kernel:
    three steps against the oracle's TF update rules, with global-norm clipping on (with these rules, unlike Adam, the
    clip scale reaches the weights).  'alternate' + SGD also covers the stale packed-weight hazard: the generator loss
    must see the discriminator that the first half of the step has just updated."""
    import argparse
    import saragan_amd.optimization as opt
    from oracle import pgan_oracle as O
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    fx = load_step_fixture(os.path.join(golden_dir, FIXTURES[3]), torch.float64)
    lr = {'SGD': 2e-3, 'Momentum': 1e-3, 'Adadelta': 1.0}[kind]
    set_compute_dtype(torch.float32)
    store = VariableStore('cuda', seed=0)
    L.set_random_source(L.InjectedRandom({k: v.float() for k, v in fx['rnd'].items()}))
    a = argparse.Namespace(optimizer=kind, d_optimizer=kind, adam_beta1=0.0, adam_beta2=0.9, d_adam_beta1=0.0,
                           d_adam_beta2=0.9, rho=0.9, d_rho=0.9, momentum=0.8, d_momentum=0.8)
    og, od = opt.get_optimizer(ScalarVariable(lr, 'd_lr'), ScalarVariable(lr, 'g_lr'), a)
    ph = opt.Placeholder([4, 1, 1, 1, 1])
    cfg = fx['cfg']
    clip = strategy == 'simultaneous'
    with use_store(store):
        tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, ScalarVariable(fx['alpha'], 'alpha'), fx['phase'],
                                BASE_SHAPE, KERNEL_SPEC, FILTER_SPEC, 'leaky_relu', 0.2, fx['loss_fn'], cfg['gp_weight'],
                                strategy, clip, clip, 0.01, None)
    store.load_state_dict(fx['p0'], strict=True)
    ema = ExtendedEMA(list(store.vars.keys()), 0.99, graph=tup[0].graph)
    sess = opt.Session('cuda')
    mk = {'SGD': O.TFSGD, 'Momentum': lambda: O.TFMomentum(0.8, True), 'Adadelta': lambda: O.TFAdadelta(0.9, 1e-7)}[kind]
    rg, rd = mk(), mk()
    p = {k: v.clone() for k, v in fx['p0'].items()}
    shadow = {k: v.clone() for k, v in p.items()}
    for step in range(3):
        _, _, gl, dl = sess.run([tup[0], tup[1], tup[2], tup[3]], feed_dict={ph: fx['real'].float()})
        sess.run(ema.apply())
        if strategy == 'simultaneous':
            ref = O.step_simultaneous(p, rg, rd, shadow, fx['rnd'], fx['real'], fx['alpha'], fx['cfg'], lr, lr,
                                      g_clipping=True, d_clipping=True)
        else:
            ref = O.step_alternate(p, rg, rd, shadow, fx['rnd'], fx['real'], fx['alpha'], fx['cfg'], lr, lr)
        np.testing.assert_allclose(float(gl), float(ref['gen_loss']), rtol=1e-4, atol=1e-5, err_msg=f'step {step}')
        np.testing.assert_allclose(float(dl), float(ref['disc_loss']), rtol=1e-4, atol=1e-5, err_msg=f'step {step}')
        # Adadelta's step is sqrt(acc_update + eps) / sqrt(acc_grad + eps) * g with eps = 1e-7 (TF default): where a
        # gradient is small against sqrt(eps) the ratio amplifies its f32 rounding (the weight-gradient sums are
        # atomic, so their last bits also vary run to run) -- 6e-5 absolute was seen on one bias element
        atol = 1e-4 if kind == 'Adadelta' else 2e-5
        for k, v in store.vars.items():
            np.testing.assert_allclose(v.detach().double().cpu().numpy(), p[k].numpy(), rtol=2e-4, atol=atol,
                                       err_msg=f'{kind} step {step} {k}')
            np.testing.assert_allclose(ema.average(k).double().cpu().numpy(), shadow[k].numpy(), rtol=2e-4, atol=atol,
                                       err_msg=f'{kind} ema step {step} {k}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_a_step_leaves_no_device_memory_in_reference_cycles(golden_dir, dtype):
    """A training step must free its activations by reference counting alone: a cycle (the fused pixel-norm stage once held
    its own output through its ActInfo) keeps device tensors alive until Python's cyclic collector happens to run -- 1.7 GiB
    per step at the benchmarked size, an out-of-memory error after a few hundred steps."""
    import gc
    fx = load_step_fixture(os.path.join(golden_dir, FIXTURES[3]), torch.float64)      # phase 3: fused pixel-norm stages
    store, tup, ph, ema, sess = _build(fx, dtype)
    feed = {ph: fx['real'].float()}
    ema_op = ema.apply()

    def step():
        sess.run([tup[0], tup[1]], feed_dict=feed)
        sess.run(ema_op)
    step()
    step()
    torch.cuda.synchronize()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        step()
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated()
        gc.set_debug(gc.DEBUG_SAVEALL)
        gc.collect()
        gc.set_debug(0)
        held = [o for o in gc.garbage if torch.is_tensor(o) and o.is_cuda]
        gc.garbage.clear()
    finally:
        if was:
            gc.enable()
    assert not held, [tuple(t.shape) for t in held]
    assert after == base, (base, after)


def test_ema_swap_is_seen_by_the_packed_weight_images(golden_dir):
    """assign_ema_weights / restore_original_weights / ema_update_weights (ExtendedEMA.py:24-59 of the reference)
    write the flat parameter buffer in place; the parameters alias it, so neither data_ptr nor the version counter the packed
    weight images are keyed by moves.  Every such writer marks the images stale itself: the generator evaluated under the EMA
    weights with FIXED latents equals an evaluation with freshly packed images, differs from the one under the training weights,
    and the restore brings that one back bit for bit."""
    from saragan_amd import functional as F
    fx = load_step_fixture(os.path.join(golden_dir, 'oracle_step_p3_wgan_a000.npz'), torch.float64)
    for dtype in (torch.float32, torch.bfloat16):
        store, tup, ph, ema, sess = _build(fx, dtype)
        train_gen, train_disc, gen_sample = tup[0], tup[1], tup[5]
        ema_op = ema.apply()
        feed = {ph: fx['real'].float()}
        for _ in range(3):      # the weights leave their shadows behind
            sess.run([train_gen, train_disc], feed_dict=feed)
            sess.run(ema_op)
        a = sess.run(gen_sample, feed_dict=feed).float().clone()            # (refreshes the images from the training weights)
        sess.run(ema.assign_ema_weights())
        b = sess.run(gen_sample, feed_dict=feed).float().clone()
        F.clear_pack_cache()
        b_ref = sess.run(gen_sample, feed_dict=feed).float().clone()
        assert torch.equal(b, b_ref), f'{dtype}: the generator under EMA weights ran on stale weight images'
        assert not torch.equal(a, b)
        sess.run(ema.restore_original_weights())
        c = sess.run(gen_sample, feed_dict=feed).float().clone()
        assert torch.equal(c, a), f'{dtype}: restore_original_weights not seen'
        sess.run(ema.ema_update_weights())
        d = sess.run(gen_sample, feed_dict=feed).float().clone()
        assert torch.equal(d, b), f'{dtype}: ema_update_weights not seen'


def test_lrmul_enters_the_packed_coefficient():
    """networks/ops.py:111-136 of the reference: a layer built with lrmul = m draws its weight with std 1 / m and multiplies it
    (and the bias) with m at run time.  Here m rides in the coefficient the weight-pack kernel applies: a layer with lrmul 2 equals the
    lrmul-1 layer whose variables hold twice the values -- output and variable gradients (which carry the factor m)."""
    from saragan_amd.networks import ops
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 4, 8, 8, generator=g).cuda().contiguous(memory_format=torch.channels_last_3d)
    outs, grads = [], []
    sa, sb = VariableStore('cuda', seed=5), VariableStore('cuda', seed=5)
    for store, m in ((sa, 2), (sb, 1)):
        with use_store(store), ops.variable_scope('layer'):
            y = ops.act(ops.apply_bias(ops.conv3d(x, 16, (3, 3, 3), 'leaky_relu', param=0.2, lrmul=m), lrmul=m), 'leaky_relu', 0.2)
            if m == 2:
                with torch.no_grad():
                    assert abs(float(store.vars['layer/weight'].std()) - 0.5) < 0.05      # init_std = 1 / lrmul
                    store.vars['layer/bias'].copy_(torch.randn(16, generator=g).cuda())
            else:
                with torch.no_grad():      # the same function with lrmul = 1: variables twice as large
                    store.vars['layer/weight'].copy_(2.0 * sa.vars['layer/weight'])
                    store.vars['layer/bias'].copy_(2.0 * sa.vars['layer/bias'])
        with use_store(store), ops.variable_scope('layer'):
            y = ops._val(ops.act(ops.apply_bias(ops.conv3d(x, 16, (3, 3, 3), 'leaky_relu', param=0.2, lrmul=m), lrmul=m), 'leaky_relu', 0.2))
        gw, gb = torch.autograd.grad(y.float().square().sum(), [store.vars['layer/weight'], store.vars['layer/bias']])
        outs.append(y.detach().float().cpu())
        grads.append((gw.detach().cpu(), gb.detach().cpu()))
    np.testing.assert_allclose(outs[0].numpy(), outs[1].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(grads[0][0].numpy(), 2.0 * grads[1][0].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(grads[0][1].numpy(), 2.0 * grads[1][1].numpy(), rtol=1e-4, atol=1e-4)
