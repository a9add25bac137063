#!/usr/bin/env python
"""bench.py -- G+D training-step throughput of the pgan hot path on MI355X (BASELINE.json metric:
3D volumes/sec/node at 128x128x32).

  python bench.py --gpus N --steps K --warmup W [--config 1..5]

N == 1 runs in this process.  N > 1: under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process
IS one rank; started plainly, bench.py spawns N fresh rank processes itself (before anything touches a GPU), waits for
them and exits non-zero unless all N came up and finished.  One rank per GPU, RCCL (torch.distributed backend "nccl").

One "step" = one pass of the hot path over one synthetic batch already resident in HBM: G forward, 4 D forwards,
gradient penalty (double backward), G and D backward, gradient all-reduce (N > 1), fused TF-Adam + EMA.
Default workload (BASELINE.json configs[2], the configuration the metric is quoted on): pgan 's' (filters
512,512,128,128,64,32), phase 6 -> volumes [n,1,32,128,128], latent 512, bf16 storage / MFMA with f32 accumulation and
f32 master weights, WGAN-GP (gp 10), stabilising phase (alpha 0), per-GPU batch fixed (weak scaling).
--config selects the other BASELINE configurations (1: xs phase 1 fp32 batch 4; 2: xs phase 4 bf16 batch 32;
4: 'm' phase 7 with fade-in, batch 2; 5: 2-D pgan 1024^2 fp32).

Rank 0 prints ONE JSON line.  At N == 1 on the default workload the same line also carries (`extras`) the fp32 rate
of the same workload and the rate with the `.npy` loader (NumpyPathDataset + PinnedPrefetcher over synthetic files)
inside the timed loop, and `cpu_baseline`: whole G+D steps of the CPU oracle timed on this host.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK = 8.0e12      # bytes/s, MI355X_MICROARCH.md
sys.path.insert(0, ROOT)

CONFIGS = {   # BASELINE.json configs[i-1]: SURVEY.md section 8d
    1: dict(size='xs', phase=1, latent=256, batch=4, dtype='f32', alpha=0.0, dims=3),
    2: dict(size='xs', phase=4, latent=256, batch=32, dtype='bf16', alpha=0.0, dims=3),
    3: dict(size='s', phase=6, latent=512, batch=32, dtype='bf16', alpha=0.0, dims=3),
    4: dict(size='m', phase=7, latent=512, batch=2, dtype='bf16', alpha=0.5, dims=3),
    5: dict(size='xs', phase=9, latent=512, batch=4, dtype='f32', alpha=0.0, dims=2),
    # the reference's OWN operating point, the only throughput it publishes (SURFGAN_3D/out.txt:18,78,84-1639: 'xs' phase 5,
    # 64x64x16, WGAN-GP 10, latent 512, LOCAL batch 2 on each of 8 Horovod ranks: 47.15 img/s global = 5.9 per GPU).
    # `--config out_txt`; --batch 4 / 8 show what the small local batches of data parallelism at 128^2 / 256^2 cost.
    6: dict(size='xs', phase=5, latent=512, batch=2, dtype='bf16', alpha=0.0, dims=3),
}
CONFIG_NAMES = {'out_txt': 6}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=lambda v: CONFIG_NAMES[v] if v in CONFIG_NAMES else int(v), default=3, choices=sorted(CONFIGS),
                    help='1..5: BASELINE.json configs[i-1]; out_txt (6): the reference log\'s own operating point')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the configuration\'s)')
    ap.add_argument('--size', default=None)
    ap.add_argument('--phase', type=int, default=None)
    ap.add_argument('--latent', type=int, default=None)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32'])
    ap.add_argument('--loss', default='wgan', choices=['wgan', 'logistic'])
    ap.add_argument('--alpha', type=float, default=None, help='0: stabilising phase; >0: mixing (freeze ops)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the fp32 and loader-in-the-loop legs')
    ap.add_argument('--dump-prof', action='store_true', help='per-shape conv kernel table on stderr')
    ap.add_argument('--cpu-budget-s', type=float, default=14.0)
    ap.add_argument('--dry-run', action='store_true',
                    help='plumbing rehearsal of the N-rank launch on CPU tensors over gloo: no GPU, no throughput (see dry_run)')
    args = ap.parse_args()
    c = CONFIGS[args.config]
    for k in ('size', 'phase', 'latent', 'batch', 'dtype', 'alpha'):
        if getattr(args, k) is None:
            setattr(args, k, c[k])
    args.dims = c['dims']
    return args


# -----------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (this process never touches a GPU)
# -----------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """The parent only counts devices and starts children; it never initialises a GPU context it would keep, and it
    never replaces its own program.  Children are polled: the first non-zero exit (a rank that died in start-up or in
    its first collective) ends the others instead of leaving them in rendezvous until the distributed timeout, and an
    overall deadline bounds the wait."""
    import socket
    import torch
    have = torch.cuda.device_count()
    stack = bool(int(os.environ.get('SARAGAN_BENCH_STACK_RANKS', '0')))    # rehearsal: several ranks share one GPU (gloo)
    if have < args.gpus and not (stack and have >= 1) and not args.dry_run:
        print(f'bench.py: --gpus {args.gpus} but only {have} device(s) are visible', file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get('SARAGAN_BENCH_DEADLINE_S', '1500'))
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed or time.time() > deadline:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            why = 'a rank failed' if failed else 'deadline passed'
            print(f'bench.py: {why}; rank exit codes {codes}', file=sys.stderr)
            return 1
        time.sleep(0.2)
    out0.seek(0)
    line = [ln for ln in out0.read().splitlines() if ln.startswith('{')]
    if not line:
        print(f'bench.py: rank 0 printed no result line; rank exit codes {codes}', file=sys.stderr)
        return 1
    rec = json.loads(line[-1])
    if rec.get('n_gpus') != args.gpus:
        print(f"bench.py: {args.gpus} ranks requested, {rec.get('n_gpus')} took part", file=sys.stderr)
        return 1
    print(line[-1], flush=True)
    return 0


# -----------------------------------------------------------------------------------------------------
# workload
# -----------------------------------------------------------------------------------------------------
def specs(args):
    if args.dims == 2:      # SURFGAN_2D: 1024^2 = 4 * 2^8 -> 9 phases (SURFGAN_2D/main.py:53), base (3,4,4), legacy triple
        from saragan_amd.networks2d.ops import num_filters
        from saragan_amd.networks2d.pgan.variables import legacy_spec
        return (3, 1, 4, 4), None, legacy_spec(9, num_filters(1, 9, size=args.size), args.size)
    from saragan_amd.networks.pgan.variables import preset_specs
    base_shape = (1, 1, 4, 4)
    ks, fs = preset_specs(args.size, base_shape, 8)
    return base_shape, ks, fs


def build(args, device, dtype):
    import torch
    import saragan_amd.optimization as opt
    from saragan_amd import parallel
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    if args.dims == 2:
        from saragan_amd.networks2d.pgan.spec_api import discriminator, generator
        from saragan_amd.networks2d.pgan.variables import variable_shapes as pgan_variable_shapes
    else:
        from saragan_amd.networks.pgan.discriminator import discriminator
        from saragan_amd.networks.pgan.generator import generator
        from saragan_amd.networks.pgan.variables import pgan_variable_shapes

    set_compute_dtype(torch.bfloat16 if dtype == 'bf16' else torch.float32)
    base_shape, ks, fs = specs(args)
    rank = parallel.rank()
    store = VariableStore(device, seed=42)            # same initial weights on every rank (then broadcast anyway)
    L.set_random_source(L.RandomSource(42 + rank, device))
    alpha = ScalarVariable(args.alpha, 'alpha')
    g_lr, d_lr = ScalarVariable(1e-3, 'g_lr'), ScalarVariable(1e-3, 'd_lr')
    og, od = opt.AdamOptimizer(g_lr, 0.0, 0.9), opt.AdamOptimizer(d_lr, 0.0, 0.9)
    if parallel.size() > 1 or (parallel.forced() and torch.distributed.is_initialized()):
        og, od = parallel.DistributedOptimizer(og), parallel.DistributedOptimizer(od)
        og.distributed.timing = od.distributed.timing = True
    sp = [s * 2 ** (args.phase - 1) for s in base_shape[1:]]
    if args.dims == 2:
        sp[0] = 1                                     # images: the D extent stays 1 (SURFGAN_2D)
    ph = opt.Placeholder([args.batch, base_shape[0], *sp])
    freeze = None
    if args.alpha > 0 and args.phase > 1:
        freeze = list(pgan_variable_shapes(args.phase - 1, base_shape, args.latent, ks, fs).keys())
    with use_store(store):
        tup = opt.optimize_step(og, od, generator, discriminator, ph, args.latent, alpha, args.phase, base_shape, ks,
                                fs, 'leaky_relu', 0.2, args.loss, 10.0 if args.loss == 'wgan' else 1.0,
                                'simultaneous', False, False, 0.01, freeze)
    graph = tup[0].graph
    ema = ExtendedEMA(list(store.vars.keys()), 0.99, graph=graph)
    graph._ensure_flat()
    parallel.broadcast_global_variables(store, 0)
    sess = opt.Session(device)
    tg, td = (tup[12], tup[16]) if freeze is not None else (tup[0], tup[1])
    return dict(store=store, sess=sess, ph=ph, train=[tg, td], ema_op=ema.apply(), ks=ks, fs=fs,
                base_shape=base_shape, shape=ph.shape, losses=[tup[3], tup[2]], graph=graph, optimizers=(og, od))


def synthetic_volume(shape, idx):
    """One LIDC-shaped synthetic sample: clip(N(1024,512),0,4095) as int16 (HU + 1024)."""
    import numpy as np
    rng = np.random.default_rng(1234 + idx)
    return np.clip(rng.normal(1024, 512, size=shape), 0, 4095).astype(np.int16)


def synthetic_batch(shape, idx, device):
    """A batch of them, normalised with mean 1024 / std 1024 (scripts/example_normal_run.jb:72), resident in HBM."""
    import numpy as np
    import torch
    v = synthetic_volume(shape, idx).astype(np.float32)
    return torch.from_numpy((v - 1024.0) / 1024.0).to(device)


def conv_flops_per_volume(ks, fs, phase, base_shape, latent, dims=3):
    """Forward conv/dense FLOPs of G and of D per volume (2*Cin*Cout*k*voxels), BASELINE.md section 2."""
    import numpy as np
    if dims == 2:
        from saragan_amd.networks2d.pgan.variables import variable_shapes as pgan_variable_shapes
    else:
        from saragan_amd.networks.pgan.variables import pgan_variable_shapes
    shapes = pgan_variable_shapes(phase, base_shape, latent, ks, fs)

    def vox(level):
        sp = [s * 2 ** (level - 1) for s in base_shape[1:]]
        if dims == 2:
            sp[0] = 1            # images are D == 1 volumes at every level: only H and W grow (SURFGAN_2D)
        return int(np.prod(sp))
    fg = fd = 0.0
    for name, shp in shapes.items():
        if not name.endswith('weight'):
            continue
        if len(shp) == 2:
            fl = 2.0 * shp[0] * shp[1]
        else:
            level = 1
            for p in name.split('/'):
                if p.startswith(('generator_block_', 'discriminator_block_', 'to_rgb_', 'from_rgb_')):
                    level = int(p.split('_')[-1])
            fl = 2.0 * np.prod(shp) * vox(level)
        if name.startswith('generator/'):
            fg += fl
        else:
            fd += fl
    return fg, fd


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


def pmc_traffic(entry, dtype):
    """Bytes per launch that crossed the L2's memory side for this (kind, shape), from the committed rocprofv3 counter
    passes (profiles/r05_pmc_traffic.json, else r04 / r03 / r02 / r01: FETCH_SIZE x2 + WRITE_SIZE; tools/pmc_probe.py +
    tools/pmc_summary.py; the counters cannot be read from inside this process).  Mean over the epilogue variants
    measured; None when this shape / batch / dtype was not part of the counter run."""
    for name in ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        tab = json.load(open(path))
        if tab.get('dtype') != dtype:
            continue
        s = entry.shape
        key = dict(n=s.n, d=s.d, h=s.h, w=s.w, cin=s.cin, cout=s.cout, k=[s.kd, s.kh, s.kw])
        kind = 'fwd' if entry.kind == 0 else 'wgrad'
        hits = [e['traffic_bytes'] for e in tab['entries'] if e['kind'] == kind and e['shape'] == key and
                e.get('variant', '').startswith(('fwd bias', 'wgrad', 'bias', 'with'))]
        if hits and not s.upsample_in:
            return round(sum(hits) / len(hits))
    return None


def sustained_mfma_peak(dtype, kernel=''):
    """TFLOP/s this board SUSTAINS on bare v_mfma_f32_32x32x16_bf16 with random operands (register-only loop, all 256 CUs,
    6 s: tools/probe/mfma_ceiling.hip, committed as profiles/r04_mfma_ceiling.txt with the clock, power and power cap
    beside it): the chip lowers its clock under MFMA load, so the 2.5 PFLOP/s spec peak is not reachable by ANY kernel on
    random data.  None for f32 (the f32 MFMA runs at the vector rate and is not clock-limited the same way) or when the
    file is not there."""
    if dtype != 'bf16':
        return None
    path = os.path.join(ROOT, 'profiles', 'r04_mfma_ceiling.txt')
    if not os.path.exists(path):
        return None
    # kernels on v_mfma_f32_16x16x32_bf16 (conv_fwd3w, conv_fwd3p16) are priced against THAT shape's ceiling: the same loop
    # sustains 2 005 TFLOP/s on it (the board holds a higher clock), 1 848 on 32x32x16
    m16 = 'conv_fwd3w' in kernel or 'conv_fwd3p16' in kernel
    for ln in open(path):
        if m16 and ln.startswith('16x16x32 random, 1 wave/SIMD'):
            return float(ln.split('last second')[1].split()[0])
        if not m16 and ln.startswith('SUSTAINED_PEAK_32x32x16_TFLOPS'):
            return float(ln.split()[1])
    return None


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


def snapshot_state(cfg):
    """Parameters (flat buffers: the variables are views of them) and optimizer state after the warm-up: every extra leg
    starts from here, as the main loop did, instead of continuing a trajectory that diverges further with every leg."""
    import torch
    flat = {p: f['param'].detach().clone() for p, f in cfg['store'].flat.items()}
    opt = [(o.t, {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in o.state.items()})
           for o in cfg['optimizers']]
    return dict(flat=flat, opt=opt)


def restore_state(cfg, snap):
    import torch
    with torch.no_grad():
        for p, t in snap['flat'].items():
            cfg['store'].flat[p]['param'].copy_(t)
        for o, (t_, st) in zip(cfg['optimizers'], snap['opt']):
            o.t = t_
            for p, d in st.items():
                for k, v in d.items():
                    if torch.is_tensor(v):
                        o.state[p][k].copy_(v)
    from saragan_amd import functional as F
    F.clear_pack_cache()


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


SETTLE_STEPS_MULTI_RANK = 60      # untimed steps after the warm-up when world > 1 (the count must match across ranks)


def dry_run(args, rank, world):
    """`--dry-run`: the plumbing of an N-rank launch exercised WITHOUT a GPU, so that the first real 8-GPU run cannot fail on
    it: spawn_ranks (or torch.distributed.run) -> rendezvous on 127.0.0.1 -> the gradient reducer's bucketed all-reduce (gloo,
    CPU tensors) inside every step -> W warm-up steps, the fixed SETTLE_STEPS_MULTI_RANK settle steps, K timed steps bracketed
    by barriers -> MAX-reduce of the ranks' durations -> ONE JSON line from rank 0.  The line says "dry_run": true and carries
    no throughput: nothing here measures anything but the launch path."""
    import torch
    from saragan_amd import parallel
    numel = 1 << 18
    param = torch.zeros(numel)
    grad = torch.zeros(numel)
    p_ = torch.nn.Parameter(param)
    p_.grad = grad
    red = parallel.GradientAllReducer(bucket_bytes=256 << 10)

    def step(i):
        grad.fill_(float(rank + 1) * (i + 1))
        red.begin(grad, [(0, numel)], [p_])
        red.finish()                                   # every bucket goes out here (no autograd hooks in the rehearsal)
        param.add_(grad, alpha=-1e-3 * red.grad_scale)

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    it = 0
    for _ in range(args.warmup):
        step(it)
        it += 1
    settle = SETTLE_STEPS_MULTI_RANK if world > 1 else 0
    for _ in range(settle):
        step(it)
        it += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(it)
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    same = True
    if torch.distributed.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        lo, hi = param.clone(), param.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    want = -1e-3 * sum(r + 1 for r in range(world)) / world * sum(range(1, it + 1))     # the averaged updates, in closed form
    ok = same and abs(float(param[0]) - want) <= 1e-4 * abs(want)
    if rank == 0:
        print(json.dumps(dict(dry_run=True, metric='plumbing rehearsal on CPU tensors (gloo): no GPU work, no throughput',
                              value=None, unit=None, n_gpus=world, steps=args.steps, warmup=args.warmup,
                              ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                              vs_baseline=None, dtype=None, data='synthetic',
                              config=dict(workload='dry run', settle=dict(steps=settle), collective=parallel.collective_info(),
                                          replicas_identical=same, update_matches_closed_form=ok))), flush=True)
    if not ok:
        raise SystemExit('dry run: the ranks disagree after the all-reduced updates')
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.dry_run:
        os.environ['SARAGAN_DIST_BACKEND'] = 'gloo'
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    import ctypes as C
    import torch
    from saragan_amd import _lib, parallel
    rank, world, local = parallel.init_distributed('gloo' if args.dry_run else None)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: every rank must be started (torch.distributed.run, '
                         f'or plain `python bench.py --gpus N`, which spawns them)')
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))   # (rehearsals may stack ranks on one GPU)
    torch.cuda.set_device(device)
    dp = torch.distributed.is_initialized()      # world > 1, or the one-rank rehearsal of the collectives (SARAGAN_DP_FORCE=1)
    comm = parallel.collective_info() if dp else None     # backend, RCCL version, communicator size, bucket algorithm
    cfg = build(args, device, args.dtype)
    sess, ph = cfg['sess'], cfg['ph']
    batches = [synthetic_batch(cfg['shape'], rank * 1000 + i, device) for i in range(4)]

    # The host enqueues a step in ~10 ms and the device runs it in ~58: unthrottled, a long run has the host hundreds of
    # steps ahead and the caching allocator cannot hand blocks back that are still queued (700 steps ended in an
    # out-of-memory error at 278 GiB).  The host waits for the step issued RUN_AHEAD steps earlier: the queue never drains,
    # so nothing changes for the device.
    RUN_AHEAD = 6
    inflight = []
    mem_diag = bool(os.environ.get('SARAGAN_BENCH_MEM'))

    def step(i):
        sess.run(cfg['train'], feed_dict={ph: batches[i % len(batches)]})
        sess.run(cfg['ema_op'])
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > RUN_AHEAD:
            inflight.pop(0).synchronize()
        if mem_diag and i % 20 == 0:
            import gc
            print(f'MEM step {i} allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved '
                  f'{torch.cuda.memory_reserved() / 2**30:.2f} GiB gc {gc.get_count()}', file=sys.stderr, flush=True)
            if i % 100 == 0 and i:
                n_ = gc.collect()
                print(f'MEM   after gc.collect() ({n_} objects): {torch.cuda.memory_allocated() / 2**30:.2f} GiB', file=sys.stderr, flush=True)

    def barrier():
        torch.cuda.synchronize()
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    snap = snapshot_state(cfg)      # the state the timed region starts from (and every extra leg: see restore_state)
    # Settle: on a fresh lease some boxes run the first seconds of sustained load ~13 % slower and then switch, between two
    # steps, to the rate every later process sees (DESIGN_NOTES.md section 5, profiles/r03_leg_windows.txt: the MFMA-bound kernels
    # take 0.75-0.83x their earlier duration, the HBM-bound ones are unchanged -- the board's state, not the program's).
    # More untimed steps, in chunks of 5 timed by HIP events, until three consecutive chunks agree to 1 % and at least 3 s
    # have passed (at most 10 s); with several ranks a fixed 60 steps (the count must match across ranks).  The model state
    # is put back afterwards.
    preheat = dict(steps=0)
    if not os.environ.get('SARAGAN_BENCH_NO_SETTLE'):
        pi = args.warmup
        if world > 1:
            for _ in range(SETTLE_STEPS_MULTI_RANK):
                step(pi)
                pi += 1
            preheat = dict(steps=SETTLE_STEPS_MULTI_RANK, rule='fixed (ranks must agree)')
        else:
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
            preheat = dict(steps=5 * len(chunk_ms), seconds=round(time.perf_counter() - t_begin, 2),
                           first_chunk_ms_per_step=round(chunk_ms[0], 3), last_chunk_ms_per_step=round(chunk_ms[-1], 3))
        # the settle steps were for the board, not for the model: the timed region trains on from the state after the W
        # warm-up steps, as it would without them (this is WGAN-GP at lr 1e-3 on noise: every extra step takes the
        # trajectory further towards the edge of the finite range)
        restore_state(cfg, snap)
    lib = _lib.load()

    def collect():
        ents = (_lib.ProfEntry * 256)()
        n_ent = C.c_int32(0)
        lib.sg_prof_collect(ents, 256, C.byref(n_ent))
        return sorted((ents[i] for i in range(n_ent.value)), key=lambda e: -e.total_ms)

    # (untimed) two steps with every conv launch bracketed by HIP events: the per-shape table and the dominant kernel.
    # Two event records per launch x ~600 launches cost ~10 % of a step, so the TIMED region below brackets only the
    # dominant (kind, shape)'s launches -- its duration is still measured inside the timed region.
    ncal = 2
    barrier()
    lib.sg_prof_enable(1)
    # The calibration steps run EAGERLY whatever the capture mode (a replayed hipGraph has no launches to bracket); the
    # environment is put back afterwards.  Unset, SARAGAN_HIPGRAPH means "capture the step if it is host-bound" (measured by
    # optimization.StepGraph on its own eager steps): the small phases replay one graph, the benchmarked one stays eager.
    env_graph = os.environ.get('SARAGAN_HIPGRAPH')
    os.environ['SARAGAN_HIPGRAPH'] = '0'
    for i in range(ncal):
        step(args.warmup + i)
    if env_graph is None:
        del os.environ['SARAGAN_HIPGRAPH']
    else:
        os.environ['SARAGAN_HIPGRAPH'] = env_graph
    step_graph = cfg['train'][0].graph
    hipgraph = any('graph' in e for e in step_graph.__dict__.get('_captures', {}).values())
    barrier()
    table = collect()
    lib.sg_prof_enable(0)
    # dominant KERNEL = the device kernel (by name) with the largest summed time over the step; the roofline object is
    # quoted on that kernel's own heaviest (kind, shape)
    by_kernel = {}
    for e in table:
        by_kernel[e.kernel] = by_kernel.get(e.kernel, 0.0) + e.total_ms
    dom_name = max(by_kernel, key=by_kernel.get) if by_kernel else None
    dom = next((e for e in table if e.kernel == dom_name), None)
    if dom is not None:
        lib.sg_prof_set_filter(dom.kind, C.byref(dom.shape))
    barrier()
    if dp:
        for o_ in cfg['optimizers']:
            o_.distributed.exposed_ms()       # forget the warm-up steps
    # (a captured step is replayed as one launch: nothing to bracket, the roofline object then quotes the calibration steps)
    lib.sg_prof_enable(0 if (os.environ.get('SARAGAN_BENCH_NO_PROF') or hipgraph) else 1)
    step_marks = [] if os.environ.get('SARAGAN_BENCH_STEP_TIMES') else None      # diagnostic: an event after every step (no sync)
    with quiet_collector(), Stopwatch(lambda: None) as sw:      # the barrier before is the one above; the one after follows
        for i in range(args.steps):
            step(args.warmup + ncal + i)
            if step_marks is not None:
                step_marks.append((torch.cuda.Event(enable_timing=True), time.perf_counter()))
                step_marks[-1][0].record()
        barrier()
    dt = sw.seconds
    if step_marks and rank == 0:
        dev_ms = [round(step_marks[i][0].elapsed_time(step_marks[i + 1][0]), 2) for i in range(len(step_marks) - 1)]
        host_ms = [round((step_marks[i + 1][1] - step_marks[i][1]) * 1e3, 2) for i in range(len(step_marks) - 1)]
        print('STEP_TIMES device', dev_ms, 'host', host_ms, file=sys.stderr, flush=True)
    if dp:      # all-reduce time left exposed behind backward, per step (G + D), this rank
        comm['exposed_allreduce_ms_per_step'] = round(sum(sum(o_.distributed.exposed_ms()) for o_ in cfg['optimizers']) / args.steps, 3)
        comm['bucket_mib'] = cfg['optimizers'][0].distributed.bucket_elems * 4 >> 20
    timed = collect()
    lib.sg_prof_enable(0)
    lib.sg_prof_set_filter(0, None)
    losses = [float(v) for v in sess.run(cfg['losses'] + cfg['train'], feed_dict={ph: batches[0]})[:2]]
    if not all(l == l and abs(l) < 1e30 for l in losses):
        raise SystemExit(f'non-finite losses after the timed steps: {losses}')
    if dp:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    rows = table
    if args.dump_prof:
        for e in rows:
            s_ = e.shape
            print(f"{'fwd ' if e.kind == 0 else 'wgrd'} {e.kernel.decode():28s} n{s_.n} {s_.d}x{s_.h}x{s_.w} {s_.cin:4d}->{s_.cout:4d} "
                  f"k{s_.kd}{s_.kh}{s_.kw} ups{s_.upsample_in} calls/step {e.launches / ncal:5.1f} "
                  f"avg {e.total_ms / e.launches * 1e3:8.1f} us ms/step {e.total_ms / ncal:7.3f} "
                  f"TF/s {e.flops_per_launch / (e.total_ms / e.launches) / 1e9:7.1f}", file=sys.stderr)
    vols = args.batch * world * args.steps
    value = vols / dt
    peak = 2500.0 if args.dtype == 'bf16' else 157.3
    roof = None
    timed = [e for e in timed if e.kernel == dom_name] or timed       # (one shape may run as several kernel variants)
    timing_note = 'HIP events around every launch of this (kernel, shape) inside the timed region'
    if hipgraph:      # the timed region replayed a hipGraph: the dominant kernel's duration comes from the eager calibration steps
        timed = sorted((e for e in table if e.kernel == dom_name and e.launches > 0), key=lambda r: -r.total_ms)
        timing_note = ('HIP events around every launch during the two eager calibration steps just before the timed region (the '
                       'timed region replays the step as ONE hipGraph: there is no launch to bracket)')
    if timed and timed[0].launches > 0:      # the dominant kernel's heaviest shape, timed inside the timed region
        best = timed[0]
        avg_ms = best.total_ms / best.launches
        ach = best.flops_per_launch / (avg_ms * 1e-3) / 1e12
        s = best.shape
        alg_bytes = int(s.n * s.d * s.h * s.w * (s.cin / (8 if s.upsample_in else 1) + s.cout) * (2 if args.dtype == 'bf16' else 4))
        # which roof bounds this (kernel, shape): its arithmetic intensity against the machine balance (peak FLOP/s over
        # 8 TB/s of HBM).  The small-channel 2-D layers of configs[4] sit below it and are priced in bytes.
        if best.flops_per_launch / alg_bytes < peak * 1e12 / HBM_PEAK:
            gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
            roof = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK / 1e9, unit='GB/s', frac=round(gbs * 1e9 / HBM_PEAK, 4))
        else:
            roof = dict(bound='mfma', achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4))
            sp = sustained_mfma_peak(args.dtype, best.kernel.decode())
            if sp:      # what the board sustains on bare MFMAs with random operands (profiles/r04_mfma_ceiling.txt)
                roof.update(sustained_peak=sp, frac_of_sustained=round(ach / sp, 4))
        roof.update(traffic=pmc_traffic(best, args.dtype), kernel=best.kernel.decode(),
                    shape=dict(n=s.n, d=s.d, h=s.h, w=s.w, cin=s.cin, cout=s.cout, k=[s.kd, s.kh, s.kw],
                               upsample_in=s.upsample_in),
                    launches=int(best.launches), avg_ms=round(avg_ms, 4), flops_per_launch=best.flops_per_launch,
                    algorithmic_bytes=alg_bytes, timing=timing_note)
        # the same kernel on its other shapes (calibration-step timings): the object above quotes the heaviest one by summed time
        others = sorted((e for e in rows if e.kernel == dom_name and e.launches > 0), key=lambda r: -r.total_ms)[:5]
        roof['by_shape'] = [dict(n=e.shape.n, cin=e.shape.cin, cout=e.shape.cout, calls_per_step=round(e.launches / ncal, 1),
                                 avg_ms=round(e.total_ms / e.launches, 4),
                                 achieved=round(e.flops_per_launch / (e.total_ms / e.launches * 1e-3) / 1e12, 1),
                                 frac=round(e.flops_per_launch / (e.total_ms / e.launches * 1e-3) / 1e12 / peak, 4)) for e in others]
    fg, fd = conv_flops_per_volume(cfg['ks'], cfg['fs'], args.phase, cfg['base_shape'], args.latent, args.dims)
    # executed conv work: G fwd+dgrad+wgrad; D: 3 forwards, 3 (wgan: the G loss reuses the D-loss data gradient)
    # or 4 data-gradient passes, 2 weight-gradient passes, 2 convs of the GP double backward + its weight gradient
    step_gf = (3 * fg + (11 if args.loss == 'wgan' else 12) * fd) / 1e9
    total_conv_ms = sum(e.total_ms for e in rows) * args.steps / ncal
    sh = cfg['shape']
    dims = 'x'.join(str(v) for v in (sh[3], sh[4], sh[2])) if args.dims == 3 else f'{sh[3]}x{sh[4]}'
    unit = 'volumes/s' if args.dims == 3 else 'images/s'
    out = dict(metric=f"3D volumes/sec/node (G+D step) at {dims}" if args.dims == 3 else f"2D images/sec/node (G+D step) at {dims}",
               value=round(value, 3), unit=unit,
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
               higher_is_better=True, scaling='weak', vs_baseline=None, dtype=args.dtype, data='synthetic',
               config=dict(workload=(f"BASELINE configs[{args.config - 1}]" if args.config <= 5 else
                                     "the reference's own operating point (SURFGAN_3D/out.txt:18,78: 47.15 img/s over 8 ranks)") +
                                    f": pgan '{args.size}' phase {args.phase} G+D step, "
                                    f"{'volumes' if args.dims == 3 else 'images'} {'x'.join(str(v) for v in sh[2:])}, "
                                    f"{args.loss}-gp, simultaneous, alpha {args.alpha}",
                           fade_branch=('pruned: alpha is exactly 0 or 1, results identical (DESIGN_NOTES.md 4.5; '
                                        'SARAGAN_NO_LERP_PRUNE=1 runs it)') if (float(args.alpha) in (0.0, 1.0) and
                                        not int(os.environ.get('SARAGAN_NO_LERP_PRUNE', '0'))) else 'computed',
                           timed_region_s=dict(wall=round(sw.wall, 4), hip_events=round(sw.gpu, 4)),
                           local_batch=args.batch, global_batch=args.batch * world, latent_dim=args.latent,
                           parallelism=f'dp{world}', collective=comm, settle=preheat, hipgraph=hipgraph, step_gflop_per_volume=round(step_gf, 1),
                           step_mfma_tflops=round(value * step_gf / 1e3 / world, 2),
                           conv_kernel_ms_per_step=round(total_conv_ms / args.steps, 3),
                           conv_ms_per_step_by_kernel={k.decode(): round(v / ncal, 3) for k, v in
                                                       sorted(by_kernel.items(), key=lambda kv: -kv[1])[:8]},
                           losses_after=dict(disc=round(losses[0], 4), gen=round(losses[1], 4))),
               roofline=roof)
    # the bandwidth-bound kernels of the step priced in bytes (calibration-step timings; configs[4]'s 4-16-channel layers): the
    # heaviest (kind, shape) of the small-channel family, algorithmic bytes = one read of x and one read / write of y
    es_ = 2 if args.dtype == 'bf16' else 4
    small = [e for e in rows if e.kernel.startswith(b'conv_small') and e.launches > 0 and      # ... those below the machine balance
             e.flops_per_launch / (e.shape.n * e.shape.d * e.shape.h * e.shape.w * (e.shape.cin + e.shape.cout) * es_) < peak * 1e12 / HBM_PEAK]
    if small:
        e = max(small, key=lambda r: r.total_ms)
        s_ = e.shape
        es = 2 if args.dtype == 'bf16' else 4
        nbytes = int(s_.n * s_.d * s_.h * s_.w * (s_.cin + s_.cout) * es)
        avg = e.total_ms / e.launches
        gbs = nbytes / (avg * 1e-3) / 1e9
        out['roofline_hbm'] = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK / 1e9, unit='GB/s', frac=round(gbs * 1e9 / HBM_PEAK, 4),
                                   traffic=pmc_traffic(e, args.dtype), kernel=e.kernel.decode(), kind='fwd' if e.kind == 0 else 'wgrad',
                                   shape=dict(n=s_.n, d=s_.d, h=s_.h, w=s_.w, cin=s_.cin, cout=s_.cout, k=[s_.kd, s_.kh, s_.kw]),
                                   launches=int(e.launches), avg_ms=round(avg, 4), algorithmic_bytes=nbytes,
                                   note='the small-channel VALU kernels (csrc/small.hip), timed during the calibration steps')
    if world == 1 and args.config == 3 and not args.no_extras:
        cpu_cfg = dict(ks=cfg['ks'], fs=cfg['fs'], base_shape=cfg['base_shape'], shape=cfg['shape'])      # (what cpu_baseline needs, kept past the legs)
        extras = {}
        try:
            restore_state(cfg, snap)      # (every leg starts where the main loop did: after the warm-up)
            extras['loader_in_loop'] = loader_leg(args, cfg, device, max(3, args.steps), barrier, snap)
            if float(args.alpha) in (0.0, 1.0):
                # the same step with the faded-out lerp branch computed as the reference's graph does (it contributes exact
                # zeros: DESIGN_NOTES.md 4.5); `value` is measured with the branch pruned
                from saragan_amd.networks import ops as _ops
                prune, _ops._NO_LERP_PRUNE = _ops._NO_LERP_PRUNE, True
                restore_state(cfg, snap)
                try:
                    for i in range(3):
                        step(i)
                    nf = max(3, args.steps // 2)
                    dtf = timed_steps(step, nf, barrier)
                    la_f = leg_losses(cfg, batches[0])
                finally:
                    _ops._NO_LERP_PRUNE = prune
                if la_f is None:
                    extras['fade_branch_computed'] = dict(value=None, invalid='the trajectory left the finite range during this leg')
                else:
                    extras['fade_branch_computed'] = dict(value=round(args.batch * nf / dtf, 3), ms_per_step=round(dtf / nf * 1e3, 3),
                                                          steps=nf, losses_after=la_f, note='alpha = 0 through sg_axpby and the previous phase\'s '
                                                          'from_rgb / to_rgb, forward and backward (SARAGAN_NO_LERP_PRUNE=1)')
            # the same workload in fp32 storage / f32-input MFMA (the reference's arithmetic, ops.py:147-150)
            del cfg, sess, batches
            from saragan_amd import functional as F
            F.clear_pack_cache()
            torch.cuda.empty_cache()
            cfg32 = build(args, device, 'f32')
            b32 = [synthetic_batch(cfg32['shape'], i, device) for i in range(2)]

            def step32(i):
                cfg32['sess'].run(cfg32['train'], feed_dict={cfg32['ph']: b32[i % 2]})
                cfg32['sess'].run(cfg32['ema_op'])
            for i in range(5):           # warm-up; then two calibration steps with every conv launch bracketed, as above
                step32(i)
            barrier()
            lib.sg_prof_enable(1)
            for i in range(ncal):
                step32(i)
            barrier()
            tab32 = collect()
            lib.sg_prof_enable(0)
            by32 = {}
            for e in tab32:
                by32[e.kernel] = by32.get(e.kernel, 0.0) + e.total_ms
            dom32 = next((e for e in tab32 if e.kernel == max(by32, key=by32.get)), None) if by32 else None
            if dom32 is not None:
                lib.sg_prof_set_filter(dom32.kind, C.byref(dom32.shape))
            lib.sg_prof_enable(1)
            n32 = max(10, args.steps // 2)
            dt32 = timed_steps(step32, n32, barrier)
            timed32 = collect()
            lib.sg_prof_enable(0)
            lib.sg_prof_set_filter(0, None)
            roof32 = None
            timed32 = [e for e in timed32 if dom32 is not None and e.kernel == dom32.kernel] or timed32
            if timed32 and timed32[0].launches > 0:
                b_ = timed32[0]
                avg32 = b_.total_ms / b_.launches
                ach32 = b_.flops_per_launch / (avg32 * 1e-3) / 1e12
                s_ = b_.shape
                roof32 = dict(bound='mfma', achieved=round(ach32, 2), peak=157.3, unit='TFLOP/s', frac=round(ach32 / 157.3, 4),
                              traffic=None, kernel=b_.kernel.decode(),
                              shape=dict(n=s_.n, d=s_.d, h=s_.h, w=s_.w, cin=s_.cin, cout=s_.cout, k=[s_.kd, s_.kh, s_.kw],
                                         upsample_in=s_.upsample_in),
                              launches=int(b_.launches), avg_ms=round(avg32, 4), flops_per_launch=b_.flops_per_launch)
            extras['f32'] = dict(value=round(args.batch * n32 / dt32, 3), ms_per_step=round(dt32 / n32 * 1e3, 3), steps=n32,
                                 warmup=5 + ncal, step_mfma_tflops=round(args.batch * n32 / dt32 * step_gf / 1e3, 2),
                                 peak_tflops=157.3, roofline=roof32,
                                 note='same workload, fp32 storage and v_mfma_f32_32x32x2_f32 (1/16 of the bf16 MFMA rate): '
                                      'the reference\'s own arithmetic (ops.py:147-150)')
        except Exception as exc:      # an extra leg must never cost the headline line
            import traceback
            traceback.print_exc()
            extras['error'] = repr(exc)[:300]
        out['extras'] = extras
        cfg = cpu_cfg
    if world == 1 and not args.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline(args, cfg, args.cpu_budget_s)
        except Exception as exc:      # (reported, never fatal for the line)
            import traceback
            traceback.print_exc()
            out['cpu_baseline'] = dict(value=None, error=repr(exc)[:300])
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
    _td = getattr(sys.modules.get('torch'), 'distributed', None)
    if _td is not None and _td.is_available() and _td.is_initialized():      # (normal completion only: every rank gets here)
        _td.destroy_process_group()
