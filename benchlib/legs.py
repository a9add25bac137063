"""bench.py: the bracket of a timed region (barrier + synchronize, wall clock and HIP events), the settle steps, the extra legs
(loader in the loop, fade branch computed, fp32) and the CPU baseline (the oracle timed on this host)."""
import shutil
import tempfile
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .launch import SETTLE_STEPS_MULTI_RANK
from .workload import build, restore_state, synthetic_batch, synthetic_volume


def cpu_baseline(args, cfg, budget_s):
    """Whole G+D optimisation steps of the CPU restatement (oracle/, fp32 torch-CPU, kind "port": TF1 cannot run here,
    SURVEY section 8c) timed on this host at batch 1 of the same workload -- forward, gradient penalty with its double
    backward, both backward passes, TF-Adam.  Bounded: steps are repeated until `budget_s` of CPU time is spent."""
    import torch
    from oracle import pgan_oracle as O
    nthreads = min(os.cpu_count() or 1, 16)   # a 1-GPU box's CPU share; more threads oversubscribe and run slower
    torch.set_num_threads(nthreads)
    ks, fs, base_shape = cfg['ks'], cfg['fs'], cfg['base_shape']
    if args.dims == 2:       # the 2-D networks as D == 1 volumes of the same restatement (oracle.specs_2d)
        ks, fs = O.specs_2d(fs['num_phases'], fs['size'])
    p = O.init_params(args.phase, base_shape, args.latent, ks, fs, seed=1, dtype=torch.float32)
    img = tuple(cfg['shape'][1:])
    ocfg = dict(phase=args.phase, base_shape=base_shape, latent_dim=args.latent, kernel_spec=ks, filter_spec=fs,
                activation='leaky_relu', leakiness=0.2, loss_fn=args.loss, gp_weight=10.0 if args.loss == 'wgan' else 1.0,
                noise_stddev=0.01)
    if args.dims == 2:
        ocfg['two_d'] = True
    nb = 1 if args.config != 1 else args.batch
    freeze = None
    if args.alpha > 0 and args.phase > 1:
        freeze = list(O.variable_shapes(args.phase - 1, base_shape, args.latent, ks, fs).keys())
    ag, ad = O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9)
    reps, t0 = 0, time.time()
    while True:
        rnd = O.draw_randomness(nb, args.latent, img, 100 + reps, dtype=torch.float32)
        real = torch.randn(nb, *img)
        O.step_simultaneous(p, ag, ad, None, rnd, real, args.alpha, ocfg, 1e-3, 1e-3, freeze=freeze)
        reps += 1
        if time.time() - t0 >= budget_s:
            break
    dt = (time.time() - t0) / reps
    return dict(value=float(nb / dt), unit='volumes/s', cores=nthreads, kind='port',
                sample=f'{reps} whole G+D step(s) of the fp32 torch-CPU oracle at batch {nb} of this workload '
                       f'({dt:.1f} s per step: G forward, 4 D forwards, GP double backward, G and D backward, TF-Adam)')


def mark(what):
    """SARAGAN_BENCH_MARK=1: wall-clock markers of the legs (tools/clock_trace.sh lines them up with rocm-smi samples)."""
    if os.environ.get('SARAGAN_BENCH_MARK'):
        print(f'MARK {time.time():.3f} {what}', flush=True)
        if what.startswith('timed region'):      # ... and a marker kernel for tools/archive/trace_windows.py (rocprofv3 --kernel-trace)
            import torch
            torch.zeros(3, device='cuda').cumsum(0)


class Stopwatch:
    """Wall time of a region bracketed by barrier + synchronize, cross-checked against a pair of HIP events on the compute
    stream.  The two agree to ~0.1 % on a healthy host; some boxes of the pool have a host clock that runs slow for seconds
    at a time (a leg of identical steps read 17 % "faster" than the kernels' own GPU time allows), so the LONGER of the
    two is the duration every rate in this file is computed from."""

    def __init__(self, barrier):
        self.barrier = barrier

    def __enter__(self):
        import torch        # not at module level: the launcher process must not initialise the GPU
        self.barrier()
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.e0.record()
        self.t0 = time.perf_counter()
        mark('timed region begins')
        return self

    def __exit__(self, *exc):
        self.e1.record()
        self.barrier()
        self.wall = time.perf_counter() - self.t0
        self.e1.synchronize()
        self.gpu = self.e0.elapsed_time(self.e1) * 1e-3
        self.seconds = max(self.wall, self.gpu)
        mark(f'timed region ends: wall {self.wall:.4f} s, events {self.gpu:.4f} s')
        return False


class quiet_collector:
    """The timed regions start with an EMPTY device queue (barrier), so a host stall in their first steps is device idle
    time: a full pass of Python's cyclic collector over the ~10^6 objects of a torch process takes ~0.1 s, and one landed
    in a 10-step region now and then (a 57.8 ms/step run read 69.5 with every kernel at its usual duration).  Collect
    first, then keep the collector off for the K steps -- what `timeit` does; a step leaves no device memory in cycles
    (tools/archive/cycle_probe.py), and the product loop freezes its long-lived objects instead (train.py)."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


def timed_steps(step, nsteps, barrier):
    with quiet_collector(), Stopwatch(barrier) as sw:
        for i in range(nsteps):
            step(i)
    return sw.seconds


def leg_losses(cfg, batch):
    """Losses of one more step after a leg: a trajectory that has left the finite range (this is WGAN at lr 1e-3 on
    noise volumes: it diverges) computes on NaNs, which the MFMA pipes run faster than on data (no operand toggling: the
    clock rises) -- such a leg's rate is not a measurement of the workload: the leg is reported as invalid (None: the
    caller drops its numbers), the headline line is still printed."""
    vals = [float(v) for v in cfg['sess'].run(cfg['losses'] + cfg['train'], feed_dict={cfg['ph']: batch})[:2]]
    if not all(v == v and abs(v) < 1e30 for v in vals):
        print(f'non-finite losses after a bench leg: {vals}', file=sys.stderr, flush=True)
        return None
    return dict(disc=round(vals[0], 4), gen=round(vals[1], 4))


def loader_leg(args, cfg, device, nsteps, barrier, snap=None):
    """The same step with the data path inside the timed loop (SURVEY section 8d "loader included"): synthetic
    `{xy}x{xy}/NNNN.npy` int16 volumes on local disk, NumpyPathDataset drawing batches, PinnedPrefetcher loading,
    normalising and copying them host-to-device on a side stream while the step runs."""
    import numpy as np
    from saragan_amd.dataset import NumpyPathDataset, PinnedPrefetcher
    shape = cfg['shape']
    tmp = tempfile.mkdtemp(prefix='saragan_bench_')
    try:
        d = os.path.join(tmp, f'{shape[-1]}x{shape[-1]}')
        os.makedirs(d)
        nfiles = max(2 * args.batch, 64)
        for i in range(nfiles):
            np.save(os.path.join(d, f'{i:04d}.npy'), synthetic_volume(tuple(shape[2:]), 10_000 + i))
        ds = NumpyPathDataset(d + '/', None, False, True, seed=42)
        pf = PinnedPrefetcher(ds, args.batch, False, mean=1024.0, stddev=1024.0, device=device, depth=2)
        sess, ph = cfg['sess'], cfg['ph']

        def step(i):
            sess.run(cfg['train'], feed_dict={ph: pf.next()})
            sess.run(cfg['ema_op'])
        # ~2 s of untimed steps: after the second or two of GPU idle spent writing the files the board's power averaging
        # lets the chip overshoot its sustained clocks, and a short leg would read up to 17 % faster than the main one
        # (DESIGN_NOTES.md section 5, profiles/r02_clock_trace.txt)
        for i in range(max(30, args.warmup + 5)):
            step(i)
        if snap is not None:      # those steps were for the board: the timed ones train on from the post-warm-up state, as the
            restore_state(cfg, snap)      # main loop's do (30 more steps of WGAN-GP at lr 1e-3 on noise left the finite range
            #                               in about one run in five, and the leg then reports no number)
        dt = timed_steps(step, nsteps, barrier)
        la = leg_losses(cfg, pf.next())
        pf.close()
        mb = nfiles * np.prod(shape[2:]) * 2 / 2 ** 20
        if la is None:
            return dict(value=None, invalid='the trajectory left the finite range during this leg')
        return dict(value=round(args.batch * nsteps / dt, 3), ms_per_step=round(dt / nsteps * 1e3, 3), steps=nsteps,
                    losses_after=la,
                    note=f'loader in the timed loop: {nfiles} synthetic int16 .npy volumes ({mb:.0f} MiB) on local disk, '
                         f'np.load -> pinned ring -> async H2D on a side stream, 2 batches ahead')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)



def settle(step, first, world):
    """Untimed steps for the BOARD: on a fresh lease some boxes run the first seconds of sustained load ~13 % slower and then
    switch, between two steps, to the rate every later process sees (DESIGN_NOTES.md section 5, profiles/r03_leg_windows.txt: the
    MFMA-bound kernels take 0.75-0.83x their earlier duration, the HBM-bound ones are unchanged -- the board's state, not the
    program's).  Chunks of 5 steps timed by HIP events, until three consecutive chunks agree to 1 % and at least 3 s have passed (at
    most 10 s); with several ranks a fixed 60 steps (the count must match across ranks).  The caller puts the model state back."""
    import torch
    if os.environ.get('SARAGAN_BENCH_NO_SETTLE'):
        return dict(steps=0)
    pi = first
    if world > 1:
        for _ in range(SETTLE_STEPS_MULTI_RANK):
            step(pi)
            pi += 1
        return dict(steps=SETTLE_STEPS_MULTI_RANK, rule='fixed (ranks must agree)')
    chunk_ms, t_begin = [], time.perf_counter()
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step(pi)
            pi += 1
        e1.record()
        torch.cuda.synchronize()
        chunk_ms.append(e0.elapsed_time(e1) / 5)
        el = time.perf_counter() - t_begin
        steady = len(chunk_ms) >= 3 and all(abs(chunk_ms[-k] - chunk_ms[-k - 1]) <= 0.01 * chunk_ms[-k] for k in (1, 2))
        if el >= 10.0 or (el >= 3.0 and steady):
            break
    return dict(steps=5 * len(chunk_ms), seconds=round(time.perf_counter() - t_begin, 2),
                first_chunk_ms_per_step=round(chunk_ms[0], 3), last_chunk_ms_per_step=round(chunk_ms[-1], 3))


def calibrate(lib, step, first, ncal, barrier):
    """(untimed) `ncal` steps with EVERY conv launch bracketed by HIP events: the per-shape table the dominant kernel is chosen from.
    Two event records per launch x ~600 launches cost ~10 % of a step, so the timed region brackets only the dominant (kind,
    shape)'s launches.  These steps run EAGERLY whatever the capture mode (a replayed hipGraph has no launches to bracket); the
    environment is put back afterwards.  Unset, SARAGAN_HIPGRAPH means "capture the step if it is host-bound" (measured by
    optimization.StepGraph on its own eager steps): the small phases replay one graph, the benchmarked one stays eager."""
    from .roofline import collect
    barrier()
    lib.sg_prof_enable(1)
    env_graph = os.environ.get('SARAGAN_HIPGRAPH')
    os.environ['SARAGAN_HIPGRAPH'] = '0'
    for i in range(ncal):
        step(first + i)
    if env_graph is None:
        del os.environ['SARAGAN_HIPGRAPH']
    else:
        os.environ['SARAGAN_HIPGRAPH'] = env_graph
    barrier()
    table = collect(lib)
    lib.sg_prof_enable(0)
    return table


def fade_leg(args, cfg, snap, step, batch, barrier):
    """The same step with the faded-out lerp branch computed as the reference's graph does (it contributes exact zeros:
    DESIGN_NOTES.md 4.5); `value` is measured with the branch pruned."""
    from saragan_amd.networks import ops as _ops
    prune, _ops._NO_LERP_PRUNE = _ops._NO_LERP_PRUNE, True
    restore_state(cfg, snap)
    try:
        for i in range(3):
            step(i)
        nf = max(3, args.steps // 2)
        dtf = timed_steps(step, nf, barrier)
        la_f = leg_losses(cfg, batch)
    finally:
        _ops._NO_LERP_PRUNE = prune
    if la_f is None:
        return dict(value=None, invalid='the trajectory left the finite range during this leg')
    return dict(value=round(args.batch * nf / dtf, 3), ms_per_step=round(dtf / nf * 1e3, 3), steps=nf, losses_after=la_f,
                note='alpha = 0 through sg_axpby and the previous phase\'s from_rgb / to_rgb, forward and backward '
                     '(SARAGAN_NO_LERP_PRUNE=1)')


def f32_leg(args, device, lib, barrier, step_gf, ncal):
    """The same workload in fp32 storage / f32-input MFMA (the reference's arithmetic, ops.py:147-150), with its own dominant
    kernel's roofline.  The caller has released the bf16 workload."""
    import ctypes as C
    from .roofline import collect, dominant, _shape_dict
    cfg32 = build(args, device, 'f32')
    b32 = [synthetic_batch(cfg32['shape'], i, device) for i in range(2)]

    def step32(i):
        cfg32['sess'].run(cfg32['train'], feed_dict={cfg32['ph']: b32[i % 2]})
        cfg32['sess'].run(cfg32['ema_op'])
    for i in range(5):           # warm-up; then two calibration steps with every conv launch bracketed, as in the main leg
        step32(i)
    barrier()
    lib.sg_prof_enable(1)
    for i in range(ncal):
        step32(i)
    barrier()
    tab32 = collect(lib)
    lib.sg_prof_enable(0)
    _, dom32, _ = dominant(tab32)
    if dom32 is not None:
        lib.sg_prof_set_filter(dom32.kind, C.byref(dom32.shape))
    lib.sg_prof_enable(1)
    n32 = max(10, args.steps // 2)
    dt32 = timed_steps(step32, n32, barrier)
    timed32 = collect(lib)
    lib.sg_prof_enable(0)
    lib.sg_prof_set_filter(0, None)
    roof32 = None
    timed32 = [e for e in timed32 if dom32 is not None and e.kernel == dom32.kernel] or timed32
    if timed32 and timed32[0].launches > 0:
        b_ = timed32[0]
        avg32 = b_.total_ms / b_.launches
        ach32 = b_.flops_per_launch / (avg32 * 1e-3) / 1e12
        roof32 = dict(bound='mfma', achieved=round(ach32, 2), peak=157.3, unit='TFLOP/s', frac=round(ach32 / 157.3, 4),
                      traffic=None, kernel=b_.kernel.decode(), shape=_shape_dict(b_.shape),
                      launches=int(b_.launches), avg_ms=round(avg32, 4), flops_per_launch=b_.flops_per_launch)
    return dict(value=round(args.batch * n32 / dt32, 3), ms_per_step=round(dt32 / n32 * 1e3, 3), steps=n32,
                warmup=5 + ncal, step_mfma_tflops=round(args.batch * n32 / dt32 * step_gf / 1e3, 2),
                peak_tflops=157.3, roofline=roof32,
                note='same workload, fp32 storage and v_mfma_f32_32x32x2_f32 (1/16 of the bf16 MFMA rate): '
                     'the reference\'s own arithmetic (ops.py:147-150)')
