"""Data-parallel layer on CPU: world_size 2 over gloo.  Covers the bucketed gradient all-reduce launched from
autograd hooks (flat buffer, several buckets, partial parameter sets as in the freeze ops) and the variable
broadcast, i.e. what replaces hvd.DistributedOptimizer / hvd.broadcast_global_variables
(optuna_objective.py:179-186,328)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from saragan_amd import parallel
    from saragan_amd.varstore import VariableStore
    r, w, _ = parallel.init_distributed('gloo')
    assert (r, w) == (rank, world) and parallel.size() == world
    torch.manual_seed(100 + rank)
    store = VariableStore('cpu', seed=7 + rank)          # different initial weights per rank
    shapes = {'generator/a/weight': (3, 5), 'generator/a/bias': (5,), 'generator/b/weight': (7, 2), 'generator/c/bias': (9,)}
    for k, s in shapes.items():
        store.get(k, s, 'normal')
    flat = store.flatten('generator/')
    parallel.broadcast_global_variables(store, 0)
    ref = VariableStore('cpu', seed=7)
    for k, s in shapes.items():
        ref.get(k, s, 'normal')
    same = all(torch.equal(store.vars[k].data, ref.vars[k].data) for k in shapes)

    red = parallel.GradientAllReducer(bucket_bytes=32)   # 8 floats per bucket: several buckets, params straddle them
    names = list(shapes)
    out = {}
    for trial, active in enumerate((names, names[2:])):  # all parameters, then a "freeze" subset
        params = [store.vars[k] for k in active]
        offs = flat['offsets']
        ranges = [(offs[k][0], (offs[k][1] + 3) // 4 * 4) for k in active]
        flat['grad'].zero_()
        for p_, k in zip(params, active):
            o, n = offs[k]
            p_.grad = flat['grad'][o:o + n].view(p_.shape)
        x = torch.full((1,), float(rank + 1))
        loss = sum((p_ * (i + 1)).sum() for i, p_ in enumerate(params)) * x
        red.begin(flat['grad'], ranges, params)
        torch.autograd.backward(loss, inputs=params)
        red.finish()
        out[trial] = {k: store.vars[k].grad.clone().numpy() for k in active}
    # a parameter the loss does not reach (the faded-out branch of a stabilising phase): begin(roots=...) counts it as
    # done, every bucket is launched by the hooks (none left to finish()), its gradient stays the cleared zeros
    params = [store.vars[k] for k in names]
    offs = flat['offsets']
    ranges = [(offs[k][0], (offs[k][1] + 3) // 4 * 4) for k in names]
    flat['grad'].zero_()
    for p_, k in zip(params, names):
        o, n = offs[k]
        p_.grad = flat['grad'][o:o + n].view(p_.shape)
    x = torch.full((1,), float(rank + 1))
    loss = sum((p_ * (i + 1)).sum() for i, p_ in enumerate(params) if names[i] != 'generator/a/bias') * x
    red.begin(flat['grad'], ranges, params, roots=[loss])
    torch.autograd.backward(loss, inputs=params)
    early = all(b['launched'] for b in red._buckets)
    red.finish()
    out[2] = {k: store.vars[k].grad.clone().numpy() for k in names}
    out['early'] = early
    q.put((rank, same, out))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_and_broadcast_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    assert all(r[1] for r in res), 'broadcast did not deliver rank 0 values'
    for trial in (0, 1):
        names = list(res[0][2][trial])
        for i, k in enumerate(names):
            want = (i + 1) * (1 + 2)                       # d/dp sum over ranks of (i+1) * p * (rank+1)
            for r in res:
                np.testing.assert_allclose(r[2][trial][k], want)
    for r in res:
        assert r[2]['early'], 'a bucket waited for finish() although its only missing parameter is unreachable'
        for i, k in enumerate(res[0][2][2]):
            np.testing.assert_allclose(r[2][2][k], 0.0 if k == 'generator/a/bias' else (i + 1) * (1 + 2))


def _adasum_numpy(vectors, segments):
    """Adasum over a binary tree of ranks, per tensor, in float64 numpy (the rule of parallel.adasum_pair)."""
    def pair(a, b):
        out = np.empty_like(a)
        for lo, hi in segments:
            x, y = a[lo:hi], b[lo:hi]
            dot, na, nb = float(x @ y), float(x @ x), float(y @ y)
            ca = 1.0 - dot / (2 * na) if na >= 1e-8 else 1.0
            cb = 1.0 - dot / (2 * nb) if nb >= 1e-8 else 1.0
            out[lo:hi] = ca * x + cb * y
        return out
    level = list(vectors)
    while len(level) > 1:
        level = [pair(level[i], level[i + 1]) for i in range(0, len(level), 2)]
    return level[0]


def _worker_algos(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from saragan_amd import parallel
    from saragan_amd.varstore import VariableStore
    parallel.init_distributed('gloo')
    info = parallel.collective_info()
    store = VariableStore('cpu', seed=3)
    shapes = {'discriminator/a/weight': (3, 5), 'discriminator/a/bias': (5,), 'discriminator/b/weight': (7, 3),
              'discriminator/c/bias': (2,)}
    for k, s in shapes.items():
        store.get(k, s, 'normal')
    flat = store.flatten('discriminator/')
    names = list(shapes)
    offs = flat['offsets']
    params = [store.vars[k] for k in names]
    ranges = [(offs[k][0], (offs[k][1] + 3) // 4 * 4) for k in names]

    def run(red, combine=False):
        flat['grad'].zero_()
        for p_, k in zip(params, names):
            o, n = offs[k]
            p_.grad = flat['grad'][o:o + n].view(p_.shape)
        g = torch.Generator().manual_seed(50 + rank)
        loss = sum((p_ * torch.randn(p_.shape, generator=g)).sum() for p_ in params)
        red.begin(flat['grad'], ranges, params)
        torch.autograd.backward(loss, inputs=params)
        red.finish()
        if combine:       # the Adasum rule itself, on the vectors the backward left in the flat buffer
            red.combine(flat['grad'])
        return flat['grad'].clone().numpy()

    out = dict(info=info)
    out['allreduce'] = run(parallel.GradientAllReducer(bucket_bytes=44))      # 11 floats: buckets that do not divide by 2 or 4
    os.environ['SARAGAN_DP_ALGO'] = 'rs_ag'
    try:
        red = parallel.GradientAllReducer(bucket_bytes=44)
        assert red.algo == 'rs_ag' and red.grad_scale == 1.0 / world
        out['rs_ag'] = run(red)
    except RuntimeError as e:           # a backend without reduce_scatter_tensor / all_gather_into_tensor
        out['rs_ag'] = f'unsupported: {e}'
    os.environ['SARAGAN_DP_ALGO'] = 'allreduce'
    os.environ['SARAGAN_DP_GRAD_DTYPE'] = 'bf16'
    try:
        red16 = parallel.GradientAllReducer(bucket_bytes=44)
        assert red16.grad_dtype == 'bf16'
        out['bf16'] = run(red16)
    except RuntimeError as e:           # a backend that cannot sum bfloat16
        out['bf16'] = f'unsupported: {e}'
    del os.environ['SARAGAN_DP_GRAD_DTYPE']
    ada = parallel.DistributedOptimizer(type('O', (), {})(), op=parallel.Adasum).distributed
    assert isinstance(ada, parallel.AdasumReducer) and ada.grad_scale == 1.0
    local = run(ada)                 # delta form: finish() leaves the gradients LOCAL (Horovod's _DistributedAdasumOptimizer)
    out['adasum_local'] = local
    out['adasum'] = run(ada, combine=True)
    # the delta form end to end with a stand-in local optimiser (p -= 0.1 * g): start + Adasum_r(p_r - start)
    run(ada)
    before = flat['param'].clone()
    lo, hi = ada.hull()
    start = flat['param'][lo:hi].clone()
    with torch.no_grad():
        for (o, n) in ranges:
            flat['param'][o:o + n] -= 0.1 * flat['grad'][o:o + n]
    ada.combine_deltas(flat['param'], start)
    out['adasum_delta'] = (flat['param'] - before).clone().numpy()
    out['segments'] = [offs[k] for k in names]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(target, world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [r[1] for r in sorted(res, key=lambda t: t[0])]


import pytest  # noqa: E402


@pytest.mark.parametrize('world', [2, 4])
def test_reduce_scatter_all_gather_form_and_adasum(world):
    """SARAGAN_DP_ALGO=rs_ag sums exactly what the all_reduce form sums (same GradientAllReducer interface, same hooks
    and buckets); the Adasum reducer (hvd.Adasum, optuna_objective.py:182-183) equals the numpy restatement of the rule
    evaluated over the binary tree of ranks, identically on every rank; collective_info names backend and size."""
    res = _run_world(_worker_algos, world)
    want_sum = None
    for r in range(world):
        g = torch.Generator().manual_seed(50 + r)
        vec = np.concatenate([np.pad(torch.randn(s, generator=g).numpy().reshape(-1), (0, (-int(np.prod(s))) % 4))
                              for s in [(3, 5), (5,), (7, 3), (2,)]])
        want_sum = vec if want_sum is None else want_sum + vec
    per_rank = []
    for r in range(world):
        g = torch.Generator().manual_seed(50 + r)
        per_rank.append(np.concatenate([np.pad(torch.randn(s, generator=g).numpy().reshape(-1), (0, (-int(np.prod(s))) % 4))
                                        for s in [(3, 5), (5,), (7, 3), (2,)]]).astype(np.float64))
    for out in res:
        assert out['info'] == dict(backend='gloo', world_size=world, algo='allreduce')
        np.testing.assert_allclose(out['allreduce'], want_sum, rtol=1e-6, atol=1e-6)
        if isinstance(out['rs_ag'], str):
            pytest.skip(out['rs_ag'])
        np.testing.assert_allclose(out['rs_ag'], want_sum, rtol=1e-6, atol=1e-6)
        # SARAGAN_DP_GRAD_DTYPE=bf16: every rank's gradient is rounded once and the partial sums are rounded as the collective
        # forms them: within (world) half-ulps of bf16 of the sum of magnitudes -- and identical on every rank
        assert not isinstance(out['bf16'], str), out['bf16']      # gloo sums bfloat16 in this torch
        if True:
            mag = np.sum([np.abs(v) for v in per_rank], axis=0)
            assert np.all(np.abs(out['bf16'] - want_sum) <= world * 2.0 ** -8 * mag + 1e-12)
            assert np.array_equal(out['bf16'], res[0]['bf16'])
            assert np.abs(out['bf16'] - want_sum).max() > 0          # it did go through bf16
        segs = [(o, o + n) for o, n in out['segments']]
        want = _adasum_numpy(per_rank, segs)
        for lo, hi in segs:
            np.testing.assert_allclose(out['adasum'][lo:hi], want[lo:hi], rtol=1e-5, atol=1e-6)
        assert np.array_equal(out['adasum'], res[0]['adasum'])
        # hvd.Adasum on a TF1 optimizer combines WEIGHT DELTAS after a local step, not gradients
        want_delta = _adasum_numpy([-0.1 * v for v in per_rank], segs)
        for lo, hi in segs:
            np.testing.assert_allclose(out['adasum_delta'][lo:hi], want_delta[lo:hi], rtol=1e-5, atol=1e-6)
        assert np.array_equal(out['adasum_delta'], res[0]['adasum_delta'])
    for r, out in enumerate(res):      # finish() reduced nothing: every rank still holds its own gradient
        np.testing.assert_allclose(out['adasum_local'], per_rank[r], rtol=1e-6, atol=1e-6)
