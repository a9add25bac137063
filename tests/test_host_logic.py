"""CPU tests of the host-side mirror: dataset sampling protocol (the reference's dataset.py:357-395 self-test),
shape/phase arithmetic, schedules, parameter-count KAT through the product's own variable enumeration, the CLI,
and that libsaragan_hip.so loads and exports every symbol include/saragan_hip.h declares."""
import os
import re

import numpy as np
import pytest

from oracle import pgan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from saragan_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'saragan_hip.h')).read()
    declared = set(re.findall(r'\b(sg_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'sg_stream_t'}
    assert len(declared) >= 23
    lib = _lib.load()                       # raises if the .so is missing or a prototype is not exported
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in saragan_hip.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes prototype'
    assert lib.sg_version().decode().startswith('saragan_hip')
    assert b'invalid' in lib.sg_error_string(-1)


def test_product_has_no_cpu_fallback():
    import torch
    from saragan_amd import functional as F
    with pytest.raises(RuntimeError, match='GPU only'):
        F.conv3d(torch.zeros(1, 8, 1, 4, 4), torch.zeros(1, 3, 3, 8, 8))
    src = ''
    for dp, _, fs in os.walk(os.path.join(ROOT, 'saragan_amd')):
        for f in fs:
            if f.endswith('.py'):
                src += open(os.path.join(dp, f)).read()
    assert 'import oracle' not in src and 'from oracle' not in src


def test_variable_plan_matches_out_txt_kat_and_oracle():
    from saragan_amd.networks.pgan.variables import pgan_variable_shapes, preset_specs
    base = (1, 1, 4, 4)
    ks, fs = preset_specs('xs', base, 8)
    oks, ofs = O.preset_specs('xs', base, 8)
    assert ks == oks and fs == ofs
    want_g = [2691585, 3872002, 4424898, 4646018, 4728994]
    want_d = [2688769, 3869441, 4422337, 4643265, 4726241]
    for phase in range(1, 6):
        shp = pgan_variable_shapes(phase, base, 512, ks, fs)
        assert dict(shp) == {k: tuple(v) for k, v in O.variable_shapes(phase, base, 512, ks, fs).items()}
        g = sum(int(np.prod(s)) for k, s in shp.items() if k.startswith('generator/'))
        d = sum(int(np.prod(s)) for k, s in shp.items() if k.startswith('discriminator/'))
        assert (g, d) == (want_g[phase - 1], want_d[phase - 1])
    with pytest.raises(ValueError):
        pgan_variable_shapes(9, base, 512, ks, fs)


def test_ops_host_functions():
    from saragan_amd.networks import ops
    assert [ops.k(x) for x in (1, 2, 3, 8)] == [1, 1, 3, 3]
    assert ops.get_kernel([4, 4, 2], [5, 3, 3]) == [3, 3, 1]
    assert ops.calculate_gain('linear') == 1
    assert ops.calculate_gain('leaky_relu', 0.2) == pytest.approx(np.sqrt(2 / 1.04))
    with pytest.raises(ValueError):
        ops.calculate_gain('leaky_relu', True)
    with pytest.raises(ValueError):
        ops.calculate_gain('gelu')
    with pytest.raises(ValueError):
        ops.act(None, 'swish')
    with pytest.raises(ValueError):
        ops.num_filters(1, 8, (1, 1, 4, 4), size='huge')
    assert [ops.num_filters(l, 8, (1, 1, 4, 4), size='m') for l in (1, 3, 6)] == [1024, 256, 64]
    a = ops.ScalarVariable(1.0, 'alpha')
    upd = ops.alpha_update(a, 64, 1.0, 8, 2)
    vals = [float(upd.run()) for _ in range(5)]
    ref, r = [], 1.0
    for _ in range(5):
        r = O.alpha_update(r, 64, 1.0, 8, 2)
        ref.append(r)
    assert vals == ref and vals[-1] == 0.0
    assert float(ops.alpha_update(ops.ScalarVariable(0.7), 0, 1.0, 8, 2).run()) == 0.0


def test_lr_update_matches_oracle_schedule():
    from saragan_amd import optimization as opt
    from saragan_amd.networks.ops import ScalarVariable
    for inc in (None, 'linear', 'exponential'):
        for dec in (None, 'linear', 'exponential'):
            lr, step = ScalarVariable(0.0), ScalarVariable(0, dtype=np.int64)
            op = opt.lr_update(lr, step, 1000, 2e-3, inc, dec, 100, 200)
            for s in (0, 50, 99, 100, 500, 800, 801, 950, 1000):
                step.assign(s)
                got = float(op.run())
                assert got == pytest.approx(O.lr_update(s, 1000, 2e-3, inc, dec, 100, 200), rel=1e-6, abs=1e-12)


def test_get_optimizer_and_optimize_step_errors():
    import argparse
    from saragan_amd import optimization as opt
    a = argparse.Namespace(optimizer='Adam', d_optimizer='SGD', adam_beta1=0.0, adam_beta2=0.9, d_adam_beta1=0.0,
                           d_adam_beta2=0.9)
    g, d = opt.get_optimizer(1e-3, 2e-3, a)
    assert isinstance(g, opt.AdamOptimizer) and isinstance(d, opt.GradientDescentOptimizer)
    a.optimizer = 'Lion'
    with pytest.raises(NotImplementedError):
        opt.get_optimizer(1e-3, 2e-3, a)
    with pytest.raises(ValueError):
        opt.optimize_step(g, d, None, None, None, 8, 0.0, 1, (1, 1, 4, 4), [], [], 'leaky_relu', 0.2, 'wgan', 10.0,
                          'sometimes', False, False, 0.01)


def test_dataset_protocol(tmp_path):
    """dataset.py:357-395: 10 files, batch(7) twice with and without auto-repeat."""
    from saragan_amd.dataset import NumpyPathDataset, invert_normalize_numpy, normalize_numpy
    d = tmp_path / '16x16'
    d.mkdir()
    for i in range(10):
        np.save(d / f'{i:03d}.npy', np.full((5, 16, 16), i, dtype=np.int16))
    ds = NumpyPathDataset(str(d) + '/', None, False, True, seed=3)
    assert len(ds) == 10 and ds.shape == (1, 5, 16, 16)
    b1 = ds.batch(7, auto_repeat=False)
    b2 = ds.batch(7, auto_repeat=False)      # only 3 left: returns what is there
    assert b1.shape == (7, 1, 5, 16, 16) and b1.dtype == np.float32 and b2.shape == (3, 1, 5, 16, 16)
    seen = sorted(int(v) for v in np.concatenate([b1, b2])[:, 0, 0, 0, 0])
    assert seen == list(range(10))           # every sample exactly once per epoch
    ds = NumpyPathDataset(str(d) + '/', None, False, True, seed=3)
    b1, b2 = ds.batch(7, True), ds.batch(7, True)   # second call refills the buffer first
    assert b1.shape == b2.shape == (7, 1, 5, 16, 16)
    tr, rest = ds.split_by_fraction(0.8)
    assert (len(tr), len(rest)) == (8, 2) and tr.scratch_files == ds.scratch_files[:8]
    x = np.arange(4.0)
    assert np.allclose(invert_normalize_numpy(normalize_numpy(x, 1024, 1024), 1024, 1024), x)
    with pytest.raises(Exception):
        normalize_numpy(x, None, 2.0)


def test_batch_mpi_partitions_the_global_batch(tmp_path):
    """The shared-seed slicing hands rank r column r of reshape(-1, world): together the ranks see each
    sample of an epoch exactly once (what rank-0 draw + MPI scatter guaranteed, dataset.py:307-333)."""
    from saragan_amd.dataset import NumpyPathDataset
    d = tmp_path / '8x8'
    d.mkdir()
    for i in range(12):
        np.save(d / f'{i:03d}.npy', np.full((2, 8, 8), i, dtype=np.int16))
    world = 3
    ranks = [NumpyPathDataset(str(d) + '/', None, False, True, rank=r, world_size=world, seed=11) for r in range(world)]
    ref = NumpyPathDataset(str(d) + '/', None, False, True, seed=11)
    epoch = []
    for _ in range(2):
        want = ref.samplebuffer[:6]
        ref.samplebuffer = ref.samplebuffer[6:]
        got = [r.batch_mpi_paths(2) for r in ranks]
        for rk in range(world):
            assert got[rk] == [want[i] for i in range(rk, 6, world)]
        epoch += [p for g_ in got for p in g_]
    assert sorted(epoch) == sorted(ref.scratch_files)
    # no auto-repeat with a short buffer: padded with None on rank 0's list, stripped per rank
    ranks = [NumpyPathDataset(str(d) + '/', None, False, True, rank=r, world_size=5, seed=1) for r in range(5)]
    for r in ranks:
        r.samplebuffer = r.samplebuffer[:7]
    got = [r.batch_mpi_paths(2, auto_repeat=False) for r in ranks]
    assert [len(g_) for g_ in got] == [2, 2, 1, 1, 1]


def test_utils_shape_math_and_log_line():
    from saragan_amd import utils as U
    assert U.get_num_phases('(1, 1, 4, 4)', '(1, 32, 128, 128)') == 5
    assert U.get_base_shape('(1, 5, 16, 16)') == (1, 5, 16, 16)
    assert U.get_current_input_shape(3, 4, '(1, 5, 16, 16)') == [4, 1, 20, 64, 64]
    assert U.get_xy_dim(6, '(1,1,4,4)') == 128
    assert U.scale_lr(1e-3, 2e-3, 'sqrt', 'linear', True, 4) == (pytest.approx(2e-3), pytest.approx(8e-3))
    assert U.scale_lr(1e-3, 2e-3, 'sqrt', 'linear', False, 4) == (1e-3, 2e-3)
    with pytest.raises(ValueError):
        U.scale_lr(1e-3, 2e-3, 'cubic', 'none', True, 2)
    line = U.format_summary_line(128, 64, 47.15, 5.9, 0.1234, -1.5, 1e-3, 2e-3, 0.5)
    assert 'Step 000000128' in line and 'img/s 47.15' in line and 'd_loss 0.1234' in line and 'alpha 0.50' in line
    assert U.get_num_metric_samples(None, 1, 8) == 16


def test_cli_flags_of_the_reference_parse():
    from saragan_amd.main import build_parser, finalize_args
    argv = ['pgan', '/data/', '--start_shape', '(1, 5, 16, 16)', '--final_shape', '(1, 160, 512, 512)', '--scratch_path',
            '/scratch', '--gpu', '--horovod', '--data_mean', '1024', '--data_stddev', '1024', '--starting_phase', '1',
            '--ending_phase', '4', '--mixing_nimg', '131072', '--stabilizing_nimg', '131072', '--base_batch_size', '16',
            '--latent_dim', '128', '--network_size', 's', '--starting_alpha', '1', '--loss_fn', 'wgan', '--gp_weight', '10',
            '--noise_stddev', '0.01', '--d_lr_increase=linear', '--d_lr_decrease=exponential', '--d_lr_rise_niter', '32768',
            '--d_lr_decay_niter', '98304', '--g_lr', '0.012250', '--d_lr', '0.005458', '--adam_beta1', '0.130724',
            '--adam_beta2', '0.939014', '--calc_metrics', '--compute_FID']   # scripts/example_normal_run.jb:70-80
    args, unknown = build_parser().parse_known_args(argv)
    args = finalize_args(args)
    assert unknown == [] and args.calc_metrics and args.compute_FID       # round 4: the metric flags are honoured (FID: warned, skipped)
    assert args.validation_fraction == 0.1 and args.test_fraction == 0.1 and args.metrics_every_nsteps == 128
    assert args.d_adam_beta1 == pytest.approx(0.130724) and args.d_optimizer == 'Adam'
    assert args.filter_spec[0] == [128, 128] and len(args.kernel_spec) >= 5   # ops.py:223-232: 5*16*16 voxels -> list index 2


def test_scratch_staging_waits_for_the_copier(tmp_path):
    """Only the node's first process copies (through a temporary name + rename); the others wait until the whole table
    is present (reference dataset.py:163-200: local_rank-0 copy and the busy-wait at :176-177)."""
    import threading
    import time
    from saragan_amd.dataset import NumpyPathDataset
    src = tmp_path / 'data' / '8x8'
    src.mkdir(parents=True)
    for i in range(6):
        np.save(src / f'{i:03d}.npy', np.full((2, 8, 8), i, dtype=np.int16))
    scratch = tmp_path / 'scratch'
    made = {}

    def waiter():
        made['waiter'] = NumpyPathDataset(str(src) + '/', str(scratch) + '/', False, True, rank=1, world_size=2, seed=5)

    th = threading.Thread(target=waiter)
    th.start()
    time.sleep(0.5)
    assert th.is_alive() and 'waiter' not in made          # nothing staged yet: still waiting
    copier = NumpyPathDataset(str(src) + '/', str(scratch) + '/', True, True, rank=0, world_size=2, seed=5)
    th.join(timeout=20)
    assert not th.is_alive()
    w = made['waiter']
    assert w.scratch_files == copier.scratch_files and len(w) == 6
    assert all(f.startswith(str(scratch)) and os.path.isfile(f) for f in w.scratch_files)
    assert not [f for f in os.listdir(w.scratch_dir) if '.part' in f]
    assert w.samplebuffer == copier.samplebuffer           # same seed, same table: same deck on every rank
    # phases other than the running one are not staged at all
    other = NumpyPathDataset(str(src) + '/', str(scratch) + '/', False, False, seed=5)
    assert other.scratch_files == other.npy_files
    with pytest.raises(TimeoutError):
        NumpyPathDataset(str(src) + '/', str(tmp_path / 'never') + '/', False, True, stage_timeout=0.3)


def test_pinned_prefetcher_host_path(tmp_path):
    """The prefetcher hands out every sample of an epoch once, normalised, through its slot ring (CPU buffers here)."""
    from saragan_amd.dataset import NumpyPathDataset, PinnedPrefetcher
    d = tmp_path / '8x8'
    d.mkdir()
    for i in range(8):
        np.save(d / f'{i:03d}.npy', np.full((2, 8, 8), 100 + i, dtype=np.int16))
    ds = NumpyPathDataset(str(d) + '/', None, False, True, seed=2)
    pf = PinnedPrefetcher(ds, 2, False, mean=100.0, stddev=2.0, device='cpu', depth=2)
    seen = []
    for _ in range(8):                                     # two epochs through a ring of four slots
        b = pf.next()
        assert tuple(b.shape) == (2, 1, 2, 8, 8) and b.dtype.is_floating_point
        seen += [float(v) for v in b[:, 0, 0, 0, 0]]
    pf.close()
    assert sorted(seen[:8]) == [i / 2.0 for i in range(8)] and sorted(seen[8:]) == sorted(seen[:8])
    ds2 = NumpyPathDataset(str(d) + '/', None, False, True, seed=2)
    bad = PinnedPrefetcher(ds2, 2, False, device='cpu')
    ds2.scratch_files[:] = [str(d / 'missing.npy')] * 8    # the worker dies: next() must say so, not hang
    with pytest.raises(RuntimeError):
        for _ in range(6):
            bad.next()
    bad.close()


def test_gradient_destination_registry():
    """functional.grads_into (host logic only): a parameter's slot of the flat gradient buffer is handed out ONCE per backward, as a
    fresh alias; later contributions find it through _grad_acc; a declined launch gives it back; nothing is handed out while a graph of
    the backward is being recorded."""
    import torch
    from saragan_amd import functional as F
    flat = torch.zeros(64)
    w = torch.nn.Parameter(torch.ones(2, 3))
    b = torch.nn.Parameter(torch.ones(3))
    slots = {w.data_ptr(): flat[8:14].view(2, 3), b.data_ptr(): flat[16:19]}
    assert F._grad_out(w.data_ptr(), (2, 3)) is None                      # no backward in flight
    with F.grads_into(slots):
        with torch.enable_grad():
            assert F._grad_out(w.data_ptr(), (2, 3)) is None              # create_graph: gradients are graph nodes, not slots
        with torch.no_grad():
            assert F._grad_acc(w.data_ptr(), (2, 3)) is None              # nothing written yet
            a = F._grad_out(w.data_ptr(), (1, 1, 1, 2, 3))
            assert a is not None and a.data_ptr() == flat[8:14].data_ptr() and a is not slots[w.data_ptr()]
            assert F._grad_out(w.data_ptr(), (2, 3)) is None              # once
            acc = F._grad_acc(w.data_ptr(), (2, 3))
            assert acc is not None and acc.data_ptr() == a.data_ptr()
            assert F._grad_out(w.data_ptr() + 4, (2, 3)) is None          # not a registered parameter
            assert F._grad_out(b.data_ptr(), (4,)) is None                # wrong size: not this parameter's gradient
            other = torch.empty(2, 3)
            F._unclaim(w.data_ptr(), other)                               # somebody else's tensor: the claim stands
            assert F._grad_acc(w.data_ptr(), (2, 3)) is not None
            F._unclaim(w.data_ptr(), a)                                   # the launch that was to write `a` declined
            assert F._grad_acc(w.data_ptr(), (2, 3)) is None and F._grad_out(w.data_ptr(), (2, 3)) is not None
            fresh = F._f32_out(b.data_ptr(), (3,), torch.device('cpu'))
            assert fresh.data_ptr() == flat[16:19].data_ptr()
            again = F._f32_out(b.data_ptr(), (3,), torch.device('cpu'))
            assert again.data_ptr() != fresh.data_ptr()                   # a second contribution gets a tensor of its own
    assert not F._GRAD_DEST


def test_autograd_adopts_a_slot_alias_as_grad():
    """The contract grads_into relies on (torch 2.x AccumulateGrad): an unset .grad adopts, without a copy, a dense gradient tensor
    nobody else holds -- also when it is a view into another buffer; a second contribution is summed out of place."""
    import torch
    flat = torch.zeros(32)
    slot = flat[4:16].view(3, 4)

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            return (x * w).sum()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            out = slot.view(3, 4)
            out.copy_(x * g)
            return None, out

    w = torch.nn.Parameter(torch.ones(3, 4))
    x = torch.arange(12.).view(3, 4)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad.data_ptr() == slot.data_ptr()))
    Fn.apply(x, w).backward()
    assert w.grad.data_ptr() == slot.data_ptr() and torch.equal(flat[4:16], x.reshape(-1)) and fired == [True]
    w.grad = None
    (Fn.apply(x, w) + (2 * w).sum()).backward()
    assert w.grad.data_ptr() != slot.data_ptr() and torch.equal(w.grad, x + 2)     # what StepGraph._land copies back


def test_mixed_in_place_and_out_of_place_contributions_are_refused():
    """a slot that took a contribution IN PLACE (functional._grad_acc) while autograd assembled the parameter's
    .grad elsewhere (a contribution in between arrived as a tensor of its own) must not be overwritten by that sum: _land raises
    instead of dropping the in-place part silently."""
    import pytest
    import torch
    from saragan_amd import functional as F
    from saragan_amd.optimization import StepGraph
    flat = torch.zeros(32)
    w = torch.nn.Parameter(torch.ones(2, 3))
    slot = flat[8:14].view(2, 3)
    slots = {id(w): slot}
    with F.grads_into({w.data_ptr(): slot}):
        with torch.no_grad():
            first = F._grad_out(w.data_ptr(), (2, 3))            # V1: written by a kernel into the slot, handed to autograd
            first.fill_(1.0)
            w.grad = first + 2.0                                  # the engine summed V1 + T2 out of place: .grad is elsewhere
            acc = F._grad_acc(w.data_ptr(), (2, 3))               # a third contribution goes into the slot in place
            acc.add_(4.0)
            F._note_accumulated(w.data_ptr())                     # (what the accumulating kernel's caller records)
            with pytest.raises(RuntimeError, match='in place'):
                StepGraph._land(w, slots)
    w.grad = None
    with F.grads_into({w.data_ptr(): slot}):                      # without the in-place part the sum is simply copied in
        with torch.no_grad():
            F._grad_out(w.data_ptr(), (2, 3)).fill_(1.0)
            w.grad = slot + 2.0
            StepGraph._land(w, slots)
    assert w.grad is slot and torch.equal(slot, torch.full((2, 3), 3.0))


def test_packed_image_cache_states():
    """functional.mark_packs_stale / clear_pack_cache (host logic only)."""
    from saragan_amd import functional as F
    F.clear_pack_cache()
    F.mark_packs_stale()
    assert F._PACK_STATE['stale'] is False          # nothing cached: nothing to refresh
    F._PACK_CACHE[('rgbmat', 1)] = (object(), object())
    F._PACK_CACHE[(123, 0, 1.0, False, 1, 8, 8, 3, 3, 3)] = (object(), object(), object(), object())
    F.mark_packs_stale()
    assert F._PACK_STATE['stale'] is True and ('rgbmat', 1) not in F._PACK_CACHE and len(F._PACK_CACHE) == 1
    F.clear_pack_cache()
    assert not F._PACK_CACHE and F._PACK_STATE['stale'] is False


def test_rs_ag_chunk_arithmetic_with_a_fake_collective(monkeypatch):
    """parallel.GradientAllReducer._launch_rs_ag at world 2 / 4 / 8 with the three collectives replaced by recorders (no process
    group): every rank's reduce-scatter output is the slice of its OWN bucket at offset rank * chunk (NCCL's documented in-place
    form), the all-gather writes the whole body from that slice, the tail (< world elements) goes through all_reduce -- and
    executing the recorded calls with their collective semantics leaves the plain sum in every rank's bucket."""
    import torch
    from saragan_amd import parallel
    calls = []

    class H:
        def wait(self):
            pass
    monkeypatch.setattr(parallel.dist, 'reduce_scatter_tensor', lambda out, inp, op=None, group=None, async_op=False: calls.append(('rs', out, inp)) or H())
    monkeypatch.setattr(parallel.dist, 'all_gather_into_tensor', lambda out, inp, group=None, async_op=False: calls.append(('ag', out, inp)) or H())
    monkeypatch.setattr(parallel.dist, 'all_reduce', lambda t, op=None, group=None, async_op=False: calls.append(('ar', t, t)) or H())
    for world in (2, 4, 8):
        for numel in (world * 5, world * 5 + 3, world - 1, 1000):
            torch.manual_seed(world * 1000 + numel)
            bufs = [torch.randn(numel) for _ in range(world)]
            want = torch.stack(bufs).sum(0)
            per_rank = []
            for r in range(world):
                red = object.__new__(parallel.GradientAllReducer)
                red.world_size, red.rank, red.group, red._ordered, red._handles = world, r, None, True, []
                chunk, body, lo, hi = red.rs_ag_plan(numel, world, r)
                assert body + (numel - body) == numel and 0 <= numel - body < world and hi - lo == chunk and hi <= body
                calls.clear()
                red._launch_rs_ag(bufs[r])
                per_rank.append(list(calls))
                kinds = [c[0] for c in calls]
                assert kinds == (['rs', 'ag'] if chunk else []) + (['ar'] if numel > body else []), (world, numel, kinds)
                if chunk:
                    _, out, inp = calls[0]
                    assert inp.data_ptr() == bufs[r].data_ptr() and inp.numel() == body
                    assert out.data_ptr() == inp.data_ptr() + r * chunk * 4 and out.numel() == chunk          # in place, NCCL's offset
                    _, gout, ginp = calls[1]
                    assert gout.data_ptr() == inp.data_ptr() and gout.numel() == body and ginp.data_ptr() == out.data_ptr()
            # execute the recorded collectives: reduce-scatter on all ranks, then the all-gathers, then the tails
            chunk = numel // world
            if chunk:
                snap = [c[0][2].clone() for c in per_rank]
                for r in range(world):
                    per_rank[r][0][1].copy_(sum(sn[r * chunk:(r + 1) * chunk] for sn in snap))
                mine = [per_rank[r][0][1].clone() for r in range(world)]
                for r in range(world):
                    per_rank[r][1][1].copy_(torch.cat(mine))
            if numel > chunk * world:
                tails = [c[-1][1].clone() for c in per_rank]
                for r in range(world):
                    per_rank[r][-1][1].copy_(sum(tails))
            for r in range(world):
                assert torch.allclose(bufs[r], want, atol=1e-5), (world, numel, r)
