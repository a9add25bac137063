"""Data parallelism on the GPU path: two ranks (gloo over CUDA tensors, both on cuda:0 so that one GPU suffices)
each run the HIP step on HALF of a golden fixture's batch with the gradient all-reduce of
parallel.DistributedOptimizer; the updated weights must equal the oracle's FULL-batch step (the losses are batch
means, so averaged half-batch gradients are the full-batch gradient): hvd.DistributedOptimizer semantics
(optuna_objective.py:179-186)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
NAME = 'oracle_step_p3_wgan_a000.npz'


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, golden, q, backend='gloo', extra_env=None):
    # gloo: both ranks share cuda:0 (one GPU suffices); nccl (= RCCL): one device per rank
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank) if backend == 'nccl' else '0', SARAGAN_DIST_BACKEND=backend,
                      HSA_ENABLE_IPC_MODE_LEGACY='0', SARAGAN_DP_TIMING='1')
    os.environ.update(extra_env or {})
    import saragan_amd.optimization as opt
    from saragan_amd import parallel
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    from tests.stepfix import BASE_SHAPE, FILTER_SPEC, KERNEL_SPEC, LATENT, load_step_fixture
    parallel.init_distributed()
    fx = load_step_fixture(os.path.join(golden, NAME), torch.float64)
    n = fx['real'].shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    set_compute_dtype(torch.float32)
    store = VariableStore('cuda', seed=100 + rank)
    L.set_random_source(L.InjectedRandom({k: v[sl].float() for k, v in fx['rnd'].items()}))
    alpha = ScalarVariable(fx['alpha'])
    og = parallel.DistributedOptimizer(opt.AdamOptimizer(ScalarVariable(1e-3), 0.0, 0.9))
    od = parallel.DistributedOptimizer(opt.AdamOptimizer(ScalarVariable(1e-3), 0.0, 0.9),
                                       op=parallel.Adasum if os.environ.get('SARAGAN_TEST_ADASUM') == '1' else parallel.Average)
    ph = opt.Placeholder([n, 1, 1, 1, 1])
    with use_store(store):
        tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, alpha, fx['phase'], BASE_SHAPE, KERNEL_SPEC,
                                FILTER_SPEC, 'leaky_relu', 0.2, fx['loss_fn'], fx['cfg']['gp_weight'], 'simultaneous',
                                False, False, 0.01, None)
    if rank == 0:
        store.load_state_dict(fx['p0'], strict=True)      # other ranks keep different weights until the broadcast
    graph = tup[0].graph
    ema = ExtendedEMA(list(store.vars), 0.99, graph=graph)
    graph._ensure_flat()
    parallel.broadcast_global_variables(store, 0)
    ema.reset_to_variables()
    sess = opt.Session('cuda')
    sess.run([tup[0], tup[1]], feed_dict={ph: fx['real'][sl].float()})
    sess.run(ema.apply())
    torch.cuda.synchronize()
    if backend == 'nccl':
        assert torch.distributed.get_backend() == 'nccl' and torch.cuda.current_device() == rank
    exposed = og.distributed.exposed_ms() + od.distributed.exposed_ms()
    assert len(exposed) == (1 if os.environ.get('SARAGAN_TEST_ADASUM') == '1' else 2) and all(e >= 0 for e in exposed)
    if extra_env and extra_env.get('SARAGAN_DP_FORCE') == '1':
        assert torch.distributed.is_initialized() and torch.distributed.get_backend() == backend
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in store.vars.items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_two_rank_step_equals_full_batch_oracle(golden_dir, backend):
    from tests.stepfix import load_step_fixture
    world = 2
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('RCCL needs one device per rank: fewer than 2 GPUs here (the 8-GPU node runs this variant)')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, golden_dir, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    fx = load_step_fixture(os.path.join(golden_dir, NAME), torch.float64)
    for k, ref in fx['p1'].items():
        for r in range(world):
            np.testing.assert_allclose(res[r][k], ref.numpy(), rtol=1e-4, atol=3e-5, err_msg=f'rank {r} {k}')
        np.testing.assert_array_equal(res[0][k], res[1][k])   # replicas stay bit-identical


@pytest.mark.parametrize('algo', ['allreduce', 'rs_ag', 'adasum'])
def test_single_rank_rccl_collectives_execute(golden_dir, algo):
    """One GPU cannot hold two RCCL ranks, but it can run every RCCL call of the N-rank path with a communicator of ONE rank
    (SARAGAN_DP_FORCE=1): start-up, the hook-launched bucket collectives on RCCL's stream against the compute stream the
    HIP kernels are enqueued on, broadcast, all three reduction forms.  The step must equal the oracle's full-batch step."""
    from tests.stepfix import load_step_fixture
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    env = dict(SARAGAN_DP_FORCE='1', SARAGAN_DP_ALGO='allreduce' if algo == 'adasum' else algo, SARAGAN_BUCKET_MIB='1',
               SARAGAN_TEST_ADASUM='1' if algo == 'adasum' else '0')
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), golden_dir, q, 'nccl', env))
    p.start()
    res = dict([q.get(timeout=300)])
    p.join(timeout=120)
    assert p.exitcode == 0
    fx = load_step_fixture(os.path.join(golden_dir, NAME), torch.float64)
    for k, ref in fx['p1'].items():
        np.testing.assert_allclose(res[0][k], ref.numpy(), rtol=1e-4, atol=3e-5, err_msg=k)


def _worker_captured(rank, world, port, golden, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      SARAGAN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import saragan_amd
    import saragan_amd.optimization as opt
    from saragan_amd import parallel
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    from tests.stepfix import BASE_SHAPE, FILTER_SPEC, KERNEL_SPEC, LATENT, load_step_fixture
    parallel.init_distributed()
    saragan_amd.set_deterministic(True)
    fx = load_step_fixture(os.path.join(golden, NAME), torch.float64)
    n = fx['real'].shape[0] // world
    set_compute_dtype(torch.float32)
    out = {}
    for mode in ('0', '1'):
        os.environ['SARAGAN_HIPGRAPH'] = mode
        store = VariableStore('cuda', seed=100)
        L.set_random_source(L.RandomSource(77 + rank, 'cuda'))
        og = parallel.DistributedOptimizer(opt.AdamOptimizer(ScalarVariable(1e-3), 0.0, 0.9))
        od = parallel.DistributedOptimizer(opt.AdamOptimizer(ScalarVariable(1e-3), 0.0, 0.9))
        ph = opt.Placeholder([n, 1, 1, 1, 1])
        with use_store(store):
            tup = opt.optimize_step(og, od, generator, discriminator, ph, LATENT, ScalarVariable(0.0), fx['phase'], BASE_SHAPE,
                                    KERNEL_SPEC, FILTER_SPEC, 'leaky_relu', 0.2, 'wgan', fx['cfg']['gp_weight'], 'simultaneous',
                                    False, False, 0.01, None)
        store.load_state_dict(fx['p0'], strict=True)
        graph = tup[0].graph
        ema = ExtendedEMA(list(store.vars), 0.99, graph=graph)
        graph._ensure_flat()
        parallel.broadcast_global_variables(store, 0)
        ema.reset_to_variables()
        sess = opt.Session('cuda')
        g = torch.Generator().manual_seed(9 + rank)
        losses = []
        graph._issue_log = []          # host-side issue order of the replayed segments and their bucket collectives
        for i in range(5):
            real = (fx['real'][rank * n:(rank + 1) * n].float() + 0.1 * torch.randn((n, *fx['real'].shape[1:]), generator=g)).cuda()
            _, _, dl = sess.run([tup[0], tup[1], tup[3]], feed_dict={ph: real})
            sess.run(ema.apply())
            losses.append(float(dl))
        torch.cuda.synchronize()
        ncap = sum(1 for e in graph.__dict__.get('_captures', {}).values() if 'graph' in e)
        out[mode] = (ncap, losses, {k: v.detach().cpu().numpy() for k, v in store.vars.items()}, list(graph._issue_log))
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_captured_step_with_a_gradient_reducer_attached(golden_dir):
    """capture is no longer switched off when a reducer is attached.  The captured region is forward +
    backward; the bucket collectives and the optimiser launches follow the replay.  Two ranks (gloo, both on cuda:0), five
    steps: bit-identical to the eager data-parallel run, replicas identical."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_captured, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        (n0, l0, w0, log0), (n1, l1, w1, log1) = res[r]['0'], res[r]['1']
        assert n0 == 0 and n1 == 1, (n0, n1)
        assert l0 == l1, (l0, l1)
        for k in w0:
            np.testing.assert_array_equal(w0[k], w1[k], err_msg=f'rank {r} {k}')
        # round 5: the step is captured as two segments ([forward + discriminator backward] [generator backward]); every replay
        # issues segment 0, the discriminator's bucket collectives, segment 1 (which overlaps them), the generator's collectives
        assert log0 == [] and len(log1) == 3 * 4, log1
        for i in range(0, len(log1), 4):
            a, b, c, d = log1[i:i + 4]
            assert a == ('segment', 0) and c == ('segment', 1), log1[i:i + 4]
            assert b[:2] == ('buckets', 0) and d[:2] == ('buckets', 1) and b[3] >= 1 and d[3] >= 1 and b[2] != d[2], log1[i:i + 4]
    for k in res[0]['1'][2]:
        np.testing.assert_array_equal(res[0]['1'][2][k], res[1]['1'][2][k])
