"""SARAGAN_HIPGRAPH=1: forward + both backward passes of the training step captured once into a hipGraph and replayed
(optimization.StepGraph); the optimizer kernels stay eager.  In reproducible mode the captured run must give BIT-IDENTICAL
weights and losses to the eager run from the same state: same kernels in the same order, the latents / mixing weights drawn
from the same generator in the same order, the instance noise from the same Philox offsets (device counter)."""
import os

import pytest
import torch

from tests.stepfix import BASE_SHAPE, FILTER_SPEC, KERNEL_SPEC, LATENT, load_step_fixture

pytestmark = pytest.mark.gpu
NAME = 'oracle_step_p3_wgan_a000.npz'


def _run(golden_dir, steps, dtype, captured):
    import saragan_amd.optimization as opt
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    fx = load_step_fixture(os.path.join(golden_dir, NAME), torch.float64)
    os.environ['SARAGAN_HIPGRAPH'] = '1' if captured else '0'
    set_compute_dtype(dtype)
    try:
        store = VariableStore('cuda', seed=0)
        L.set_random_source(L.RandomSource(1234, 'cuda'))
        og = opt.AdamOptimizer(ScalarVariable(1e-3, 'g_lr'), 0.0, 0.9)
        od = opt.AdamOptimizer(ScalarVariable(1e-3, 'd_lr'), 0.0, 0.9)
        ph = opt.Placeholder([4, 1, 1, 1, 1])
        with use_store(store):
            tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, ScalarVariable(0.0, 'alpha'), fx['phase'],
                                    BASE_SHAPE, KERNEL_SPEC, FILTER_SPEC, 'leaky_relu', 0.2, 'wgan', fx['cfg']['gp_weight'],
                                    'simultaneous', False, False, 0.01, None)
        store.load_state_dict(dict(fx['p0']), strict=True)
        ema = ExtendedEMA(list(store.vars.keys()), 0.99, graph=tup[0].graph)
        ema_op = ema.apply()
        sess = opt.Session('cuda')
        g = torch.Generator().manual_seed(5)
        losses = []
        for i in range(steps):
            real = (fx['real'].float() + 0.1 * torch.randn(fx['real'].shape, generator=g)).cuda()
            _, _, gl, dl = sess.run([tup[0], tup[1], tup[2], tup[3]], feed_dict={ph: real})
            sess.run(ema_op)
            losses.append((float(gl), float(dl)))
        graph = tup[0].graph
        ncap = sum(1 for e in graph.__dict__.get('_captures', {}).values() if 'graph' in e)
        return {k: v.detach().clone() for k, v in store.vars.items()}, losses, ncap
    finally:
        os.environ['SARAGAN_HIPGRAPH'] = '0'
        set_compute_dtype(torch.float32)
        L.set_random_source(None)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_captured_step_equals_eager_step_bit_for_bit(golden_dir, dtype):
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, n0 = _run(golden_dir, 6, dtype, captured=False)
        w1, l1, n1 = _run(golden_dir, 6, dtype, captured=True)
    finally:
        saragan_amd.set_deterministic(False)
    assert n0 == 0 and n1 == 1                  # two eager warm-up steps, one capture, replays after
    assert l0 == l1, (l0, l1)
    bad = [k for k in w0 if not torch.equal(w0[k], w1[k])]
    assert not bad, bad
    assert all(abs(v) < 1e6 for pair in l0 for v in pair)
