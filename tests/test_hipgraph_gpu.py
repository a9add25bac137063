"""The training step captured once into a hipGraph and replayed (optimization.StepGraph): forward, both backward passes, the
global-norm clip and the optimiser + EMA launches.  In reproducible mode the captured run must give BIT-IDENTICAL weights and
losses to the eager run from the same state: same kernels in the same order, the latents / mixing weights drawn from the same
generator in the same order, the instance noise from the same Philox offsets (device counter), the fade-in weights and the
optimisers' step sizes read from device scalars the host refreshes before each replay (round 4: a MIXING phase, whose alpha
moves every step, and a learning-rate schedule replay one graph).  SARAGAN_HIPGRAPH=1 forces capture, =0 forbids it, unset
the step is captured when it turns out host-bound -- which these toy steps are."""
import os

import numpy as np
import pytest
import torch

from tests.stepfix import BASE_SHAPE, FILTER_SPEC, KERNEL_SPEC, LATENT, load_step_fixture

pytestmark = pytest.mark.gpu
NAME = 'oracle_step_p3_wgan_a000.npz'


def _run(golden_dir, steps, dtype, captured, mixing=False, clipping=False, keep=False, reload_at=None, run_ahead=False, alpha_step=0.11, strategy='simultaneous'):
    import saragan_amd.optimization as opt
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    fx = load_step_fixture(os.path.join(golden_dir, NAME), torch.float64)
    if captured == 'auto':
        os.environ.pop('SARAGAN_HIPGRAPH', None)
    else:
        os.environ['SARAGAN_HIPGRAPH'] = '1' if captured else '0'
    set_compute_dtype(dtype)
    try:
        store = VariableStore('cuda', seed=0)
        L.set_random_source(L.RandomSource(1234, 'cuda'))
        g_lr, d_lr = ScalarVariable(1e-3, 'g_lr'), ScalarVariable(1e-3, 'd_lr')
        og = opt.AdamOptimizer(g_lr, 0.0, 0.9)
        od = opt.AdamOptimizer(d_lr, 0.0, 0.9)
        ph = opt.Placeholder([4, 1, 1, 1, 1])
        alpha = ScalarVariable(0.95 if mixing else 0.0, 'alpha')
        freeze = None
        if mixing:      # quirk Q4: the previous phase's variables stay frozen while alpha > 0
            from oracle import pgan_oracle as O
            freeze = list(O.variable_shapes(fx['phase'] - 1, BASE_SHAPE, LATENT, KERNEL_SPEC, FILTER_SPEC).keys())
        with use_store(store):
            tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, alpha, fx['phase'],
                                    BASE_SHAPE, KERNEL_SPEC, FILTER_SPEC, 'leaky_relu', 0.2, 'wgan', fx['cfg']['gp_weight'],
                                    strategy, clipping, clipping, 0.01, freeze)
        store.load_state_dict(dict(fx['p0']), strict=True)
        ema = ExtendedEMA(list(store.vars.keys()), 0.99, graph=tup[0].graph)
        ema_op = ema.apply()
        sess = opt.Session('cuda')
        g = torch.Generator().manual_seed(5)
        losses, kept = [], []
        tg, td = (tup[12], tup[16]) if mixing else (tup[0], tup[1])
        fetch = [tg, td, tup[2], tup[3]] + ([tup[10], tup[11]] if clipping and not mixing else [])
        reals = [(fx['real'].float() + 0.1 * torch.randn(fx['real'].shape, generator=g)).cuda() for _ in range(steps)]
        if run_ahead:      # nothing in the loop below waits for the device; a long launch up front lets the host get steps ahead
            big = torch.randn(8192, 8192, device='cuda')
            for _ in range(12):
                big = (big @ big).clamp_(-1, 1)
        for i in range(steps):
            g_lr.assign(1e-3 * (1.0 + 0.1 * i))       # a schedule: the step size moves every step
            d_lr.assign(1e-3 * (1.0 - 0.05 * i) if not run_ahead else 1e-3 / (1.0 + 0.05 * i))
            real = reals[i]
            if i == reload_at:      # the weights change under the graph (a checkpoint restored mid-run): torch copies, new versions
                store.load_state_dict({k: v * 0.5 for k, v in fx['p0'].items()}, strict=True)
            res = sess.run(fetch, feed_dict={ph: real})
            gl, dl = res[2], res[3]
            sess.run(ema_op)
            if keep:
                kept.append((gl, dl))                 # read only after the loop: a fetched tensor is the caller's to keep
            else:
                losses.append((float(gl), float(dl)) + tuple(float(v) for v in res[4:]))
            if mixing:
                alpha.assign(max(float(alpha.eval()) - alpha_step, 0.0))
        if keep:
            losses = [(float(a), float(b)) for a, b in kept]
        graph = tup[0].graph
        ncap = sum(1 for e in graph.__dict__.get('_captures', {}).values() if 'graph' in e)
        state = {k: v.detach().clone() for k, v in store.vars.items()}
        state.update({'ema/' + p: ema.shadow_flat(p).clone() for p in ('generator/', 'discriminator/')})
        return state, losses, ncap
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


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_captured_mixing_phase_replays_one_graph(golden_dir, dtype):
    """alpha moves every step (networks/ops.py:4-23) and the freeze train ops run (quirk Q4): ONE captured graph serves the
    whole phase -- alpha and 1 - alpha are device scalars -- and equals the eager run bit for bit (weights, EMA shadows, losses)."""
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, n0 = _run(golden_dir, 7, dtype, captured=False, mixing=True)
        w1, l1, n1 = _run(golden_dir, 7, dtype, captured=True, mixing=True)
    finally:
        saragan_amd.set_deterministic(False)
    assert n0 == 0 and n1 == 1, (n0, n1)
    assert l0 == l1, (l0, l1)
    assert len({a for a, _ in l0}) == len(l0)   # the losses do move with alpha: the replays did not reuse a stale weight
    bad = [k for k in w0 if not torch.equal(w0[k], w1[k])]
    assert not bad, bad


def test_replays_that_run_ahead_of_the_device_read_their_own_scalars(golden_dir):
    """a replayed step costs the host almost nothing, so without a synchronisation in the loop (train.py only
    synchronises when it logs, on rank 0) the host is many steps ahead of the device.  The fade-in weights and step sizes travel
    through a pinned mirror and an asynchronous copy: rewriting the mirror before that copy has executed hands a replay the
    scalars of a LATER step.  30 mixing steps queued behind a long launch, no fetch read before the end: bit-identical to eager."""
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, n0 = _run(golden_dir, 30, torch.float32, captured=False, mixing=True, keep=True, run_ahead=True, alpha_step=0.03)
        w1, l1, n1 = _run(golden_dir, 30, torch.float32, captured=True, mixing=True, keep=True, run_ahead=True, alpha_step=0.03)
    finally:
        saragan_amd.set_deterministic(False)
    assert n0 == 0 and n1 == 1, (n0, n1)
    assert l0 == l1, [(i, a, b) for i, (a, b) in enumerate(zip(l0, l1)) if a != b][:4]
    bad = [k for k in w0 if not torch.equal(w0[k], w1[k])]
    assert not bad, bad


@pytest.mark.parametrize('mixing', [False, True])
def test_captured_alternate_step_equals_eager(golden_dir, mixing):
    """--optim_strategy alternate (optimization.py:166-216 of the reference: D step, then the generator loss on the UPDATED
    discriminator) captured as one graph: the second half-step's first convolution refreshes the weight images from the weights the
    first half-step's optimiser launch wrote, the two forward passes draw their own latents (StaticRandom's pattern z, g, z).
    Bit-identical to the eager run, stabilising and mixing (freeze train ops, alpha and the step sizes moving every step)."""
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, n0 = _run(golden_dir, 6, torch.float32, captured=False, mixing=mixing, strategy='alternate')
        w1, l1, n1 = _run(golden_dir, 6, torch.float32, captured=True, mixing=mixing, strategy='alternate')
    finally:
        saragan_amd.set_deterministic(False)
    assert n0 == 0 and n1 == 1, (n0, n1)
    assert l0 == l1, (l0, l1)
    bad = [k for k in w0 if not torch.equal(w0[k], w1[k])]
    assert not bad, bad


def test_captured_step_with_clipping_norm_fetches_and_kept_outputs(golden_dir):
    """Global-norm clipping and the max-norm fetches inside the graph; fetched tensors stay valid after later replays
    (the replay path used to hand out aliases of the graph's static output buffers)."""
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, _ = _run(golden_dir, 6, torch.float32, captured=False, clipping=True)
        w1, l1, n1 = _run(golden_dir, 6, torch.float32, captured=True, clipping=True)
        _, l2, _ = _run(golden_dir, 6, torch.float32, captured=True, keep=True)
        _, l3, _ = _run(golden_dir, 6, torch.float32, captured=False)
    finally:
        saragan_amd.set_deterministic(False)
    assert n1 == 1
    assert l0 == l1, (l0, l1)
    assert not [k for k in w0 if not torch.equal(w0[k], w1[k])]
    assert l2 == [v[:2] for v in l3], (l2, l3)


def test_host_bound_step_is_captured_without_a_switch(golden_dir):
    """No SARAGAN_HIPGRAPH in the environment: the toy step (a few hundred 10-microsecond kernels behind Python) measures
    host-bound on its eager steps 2 and 3 and is captured; same results as the eager run."""
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, _ = _run(golden_dir, 8, torch.float32, captured=False)
        w1, l1, n1 = _run(golden_dir, 8, torch.float32, captured='auto')
    finally:
        saragan_amd.set_deterministic(False)
    assert n1 == 1, n1
    assert l0 == l1
    assert not [k for k in w0 if not torch.equal(w0[k], w1[k])]


def test_capture_refuses_a_live_autograd_graph():
    """the invariant behind round 3's hipStreamEndCapture crash is checked, not assumed (CPU-visible state)."""
    import saragan_amd.optimization as opt
    ps = [torch.nn.Parameter(torch.randn(4, device='cuda')) for _ in range(3)]
    opt.StepGraph.assert_no_live_accumulate_grad(ps)
    y = (ps[2] * 2).sum()
    with pytest.raises(RuntimeError, match='AccumulateGrad'):
        opt.StepGraph.assert_no_live_accumulate_grad(ps)
    del y
    opt.StepGraph.assert_no_live_accumulate_grad(ps)


def test_a_capture_that_cannot_start_raises_and_leaves_the_run_usable(golden_dir, monkeypatch):
    """Somebody keeps the previous step's autograd graph alive (here: a spy that stores the undetached loss): the capture must
    not begin -- a Python error, not a crash in hipStreamEndCapture --, the optimisers' step counts are put back, and the key
    stays eager afterwards."""
    import saragan_amd.optimization as opt
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    fx = load_step_fixture(os.path.join(golden_dir, NAME), torch.float64)
    monkeypatch.setenv('SARAGAN_HIPGRAPH', '1')
    set_compute_dtype(torch.float32)
    kept = []
    real_compute = opt.StepGraph._compute_simultaneous

    def spy(self, real, train_ids, net_args, out, arm_dist=True):
        pend = real_compute(self, real, train_ids, net_args, out, arm_dist=arm_dist)
        kept[:] = [out['disc_loss']]            # holds the graph of the step that just ran
        return pend
    monkeypatch.setattr(opt.StepGraph, '_compute_simultaneous', spy)
    try:
        store = VariableStore('cuda', seed=0)
        L.set_random_source(L.RandomSource(99, 'cuda'))
        og = opt.AdamOptimizer(ScalarVariable(1e-3, 'g_lr'), 0.0, 0.9)
        od = opt.AdamOptimizer(ScalarVariable(1e-3, 'd_lr'), 0.0, 0.9)
        ph = opt.Placeholder([4, 1, 1, 1, 1])
        with use_store(store):
            tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, ScalarVariable(0.0, 'alpha'), fx['phase'],
                                    BASE_SHAPE, KERNEL_SPEC, FILTER_SPEC, 'leaky_relu', 0.2, 'wgan', fx['cfg']['gp_weight'],
                                    'simultaneous', False, False, 0.01, None)
        store.load_state_dict(dict(fx['p0']), strict=True)
        ExtendedEMA(list(store.vars.keys()), 0.99, graph=tup[0].graph)
        sess = opt.Session('cuda')
        real = fx['real'].float().cuda()
        for _ in range(2):                        # the two eager warm-up steps of the key
            sess.run([tup[0], tup[1]], feed_dict={ph: real})
        assert (og.t, od.t) == (2, 2)
        with pytest.raises(RuntimeError, match='AccumulateGrad'):
            sess.run([tup[0], tup[1]], feed_dict={ph: real})
        assert (og.t, od.t) == (2, 2)             # nothing was applied, nothing was counted
        kept.clear()
        _, _, dl = sess.run([tup[0], tup[1], tup[3]], feed_dict={ph: real})      # the key went back to the eager path
        assert (og.t, od.t) == (3, 3) and np.isfinite(float(dl))
        assert not any('graph' in e for e in tup[0].graph.__dict__['_captures'].values())
    finally:
        set_compute_dtype(torch.float32)
        L.set_random_source(None)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gradients_written_straight_into_the_flat_buffer(golden_dir, dtype, monkeypatch):
    """F.grads_into: the first gradient of every parameter is written by its kernel into the parameter's slot of the flat
    gradient buffer and adopted by autograd as .grad; only sums of several contributions (the discriminator's weights under the
    gradient penalty) are copied in.  Same weights and losses, bit for bit, as with every gradient a tensor of its own that
    autograd adds onto the zeroed buffer (the round-3 path, SARAGAN_NO_GRAD_DEST=1)."""
    import saragan_amd
    from saragan_amd import functional as F
    saragan_amd.set_deterministic(True)
    try:
        for k in F.GRAD_DEST_STATS:
            F.GRAD_DEST_STATS[k] = 0
        w1, l1, _ = _run(golden_dir, 4, dtype, captured=False)
        st = dict(F.GRAD_DEST_STATS)
        monkeypatch.setattr(F, '_NO_GRAD_DEST', True)
        for k in F.GRAD_DEST_STATS:
            F.GRAD_DEST_STATS[k] = 0
        w0, l0, _ = _run(golden_dir, 4, dtype, captured=False)
        st0 = dict(F.GRAD_DEST_STATS)
    finally:
        saragan_amd.set_deterministic(False)
    assert l0 == l1
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
    print(st, st0)
    # phase 3: 4 steps x (generator backward + discriminator backward); every reached parameter's gradient was claimed and adopted
    # in place; the discriminator's filters' second contribution (gradient penalty) was added by the weight-gradient kernels
    # (left to a copy: layers on the pointwise kernels with two contributions -- from_rgb and the last dense layer, 1 channel wide)
    assert st['claimed'] > 0 and st['accumulated'] > 0 and st['adopted'] + st['copied'] == st['claimed'] and st['copied'] <= 2 * 4, st
    assert st0['claimed'] == 0 and st0['adopted'] == 0 and st0['accumulated'] == 0 and st0['copied'] == st['claimed'], (st, st0)


@pytest.mark.parametrize('captured', [False, True])
def test_weight_images_refreshed_in_one_launch_per_step(golden_dir, captured, monkeypatch):
    """After an optimiser step every layer's packed weight image is stale at once: the images stay where they are and ONE
    sg_conv3d_pack_weights_batch launch rewrites them when the next convolution asks for one (eager), or as the first node of the
    captured step.  Same weights and losses, bit for bit, as with the images dropped and packed one by one on use
    (SARAGAN_NO_PACK_BATCH=1, the round-3 behaviour)."""
    import saragan_amd
    from saragan_amd import functional as F
    saragan_amd.set_deterministic(True)
    try:
        F.clear_pack_cache()
        for k in F.PACK_STATS:
            F.PACK_STATS[k] = 0
        w1, l1, n1 = _run(golden_dir, 6, torch.bfloat16, captured=captured)
        st = dict(F.PACK_STATS)
        monkeypatch.setattr(F, '_NO_PACK_BATCH', True)
        F.clear_pack_cache()
        for k in F.PACK_STATS:
            F.PACK_STATS[k] = 0
        w0, l0, n0 = _run(golden_dir, 6, torch.bfloat16, captured=captured)
        st0 = dict(F.PACK_STATS)
    finally:
        saragan_amd.set_deterministic(False)
        F.clear_pack_cache()
    print(st, st0)
    assert l0 == l1 and n0 == n1 == (1 if captured else 0)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
    assert st0['batches'] == 0 and st['batches'] >= (1 if captured else 5)
    assert st['single'] < st0['single'] / 2


def test_weights_restored_between_replays_are_seen_by_the_captured_step(golden_dir):
    """The captured step rewrites its packed weight images from the weights of the day as its first node: weights that are
    replaced between two replays (load_state_dict: a checkpoint restored mid-run) are what the next replay computes with --
    bit-identical to the eager run that does the same."""
    import saragan_amd
    saragan_amd.set_deterministic(True)
    try:
        w0, l0, n0 = _run(golden_dir, 7, torch.bfloat16, captured=False, reload_at=4)
        w1, l1, n1 = _run(golden_dir, 7, torch.bfloat16, captured=True, reload_at=4)
    finally:
        saragan_amd.set_deterministic(False)
    assert n0 == 0 and n1 == 1
    assert l0 == l1
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
