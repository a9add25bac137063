#!/usr/bin/env python
"""bench.py -- G+D training-step throughput of the pgan hot path on MI355X (BASELINE.json metric:
3D volumes/sec/node at 128x128x32).

  python bench.py --gpus N --steps K --warmup W         (N == 1: run directly; N > 1: under torch.distributed.run)

One "step" = one pass of the hot path over one synthetic batch already resident in HBM: G forward, 4 D forwards,
gradient penalty (double backward), G and D backward, gradient all-reduce (N > 1), fused TF-Adam + EMA.
Workload at every N: pgan 's' (filters 512,512,128,128,64,32), phase 6 -> volumes [n,1,32,128,128], latent 512,
bf16 storage/MFMA with f32 accumulation and f32 master weights, WGAN-GP (gp 10), stabilising phase (alpha 0),
per-GPU batch fixed (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--size', default='s')
    ap.add_argument('--phase', type=int, default=6)
    ap.add_argument('--latent', type=int, default=512)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--loss', default='wgan', choices=['wgan', 'logistic'])
    ap.add_argument('--alpha', type=float, default=0.0, help='0: stabilising phase; >0: mixing (freeze ops)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-prof', action='store_true', help='per-shape conv kernel table on stderr')
    ap.add_argument('--cpu-budget-s', type=float, default=30.0)
    return ap.parse_args()


def build(args, device):
    import saragan_amd.optimization as opt
    from saragan_amd import parallel
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.networks.pgan.variables import pgan_variable_shapes, preset_specs
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store

    set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    base_shape = (1, 1, 4, 4)
    ks, fs = preset_specs(args.size, base_shape, 8)
    rank = parallel.rank()
    store = VariableStore(device, seed=42)            # same initial weights on every rank (then broadcast anyway)
    L.set_random_source(L.RandomSource(42 + rank, device))
    alpha = ScalarVariable(args.alpha, 'alpha')
    g_lr, d_lr = ScalarVariable(1e-3, 'g_lr'), ScalarVariable(1e-3, 'd_lr')
    og, od = opt.AdamOptimizer(g_lr, 0.0, 0.9), opt.AdamOptimizer(d_lr, 0.0, 0.9)
    if parallel.size() > 1:
        og, od = parallel.DistributedOptimizer(og), parallel.DistributedOptimizer(od)
    sp = [s * 2 ** (args.phase - 1) for s in base_shape[1:]]
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
                base_shape=base_shape, shape=ph.shape, losses=[tup[3], tup[2]])


def synthetic_batch(shape, idx, device):
    """LIDC-shaped synthetic volumes: clip(N(1024,512),0,4095) int16, normalised with
    mean 1024 / std 1024, resident in HBM."""
    rng = np.random.default_rng(1234 + idx)
    v = np.clip(rng.normal(1024, 512, size=shape), 0, 4095).astype(np.int16).astype(np.float32)
    return torch.from_numpy((v - 1024.0) / 1024.0).to(device)


def conv_flops_per_volume(ks, fs, phase, base_shape, latent):
    """Forward conv/dense FLOPs of G and of D per volume (2*Cin*Cout*k*voxels), BASELINE.md section 2."""
    from saragan_amd.networks.pgan.variables import pgan_variable_shapes
    shapes = pgan_variable_shapes(phase, base_shape, latent, ks, fs)

    def vox(level):
        return int(np.prod([s * 2 ** (level - 1) for s in base_shape[1:]]))
    fg = fd = 0.0
    for name, shp in shapes.items():
        if not name.endswith('weight'):
            continue
        if len(shp) == 2:
            fl = 2.0 * shp[0] * shp[1]
        else:
            parts = name.split('/')
            level = 1
            for p in parts:
                if p.startswith('generator_block_') or p.startswith('discriminator_block_'):
                    level = int(p.split('_')[-1])
                if p.startswith('to_rgb_') or p.startswith('from_rgb_'):
                    level = int(p.split('_')[-1])
            fl = 2.0 * np.prod(shp) * vox(level)
        if name.startswith('generator/'):
            fg += fl
        else:
            fd += fl
    return fg, fd


def cpu_baseline(args, cfg, budget_s):
    """The CPU restatement (oracle/, fp32 torch-CPU, kind "port") timed on this host on a BOUNDED sample of the same
    workload: one discriminator forward pass over one volume (a full G+D step of this network takes ~5 minutes on
    256 host threads).  D-forward is F_D of the step's 3*F_G + 12*F_D algorithmic FLOPs; the step rate is that
    time scaled by the FLOP ratio (backward passes are not faster than forward on the CPU, so this favours the CPU)."""
    from oracle import pgan_oracle as O
    nthreads = min(os.cpu_count() or 1, 16)   # a 1-GPU box's CPU share; more threads oversubscribe and run slower
    torch.set_num_threads(nthreads)
    ks, fs, base_shape = cfg['ks'], cfg['fs'], cfg['base_shape']
    p = O.init_params(args.phase, base_shape, args.latent, ks, fs, seed=1, dtype=torch.float32)
    img = tuple(cfg['shape'][1:])
    real = torch.randn(1, *img)
    fg, fd = conv_flops_per_volume(ks, fs, args.phase, base_shape, args.latent)
    ratio = (3 * fg + 12 * fd) / fd
    reps, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            O.discriminator(p, real, args.alpha, args.phase, args.latent, 'leaky_relu', ks, fs, param=0.2)
            reps += 1
            if time.time() - t0 > 12.0 and reps >= 3:   # ~12 s of CPU work
                break
    dt = (time.time() - t0) / reps
    return dict(value=float(1.0 / (dt * ratio)), unit='volumes/s', cores=nthreads, kind='port',
                sample=f'{reps} discriminator forward pass(es) over one {img[1]}x{img[2]}x{img[3]} volume '
                       f'({dt:.1f} s each, fp32 torch-CPU restatement in oracle/), scaled by the step/forward '
                       f'FLOP ratio {ratio:.1f}')


def pmc_traffic(entry, dtype):
    """Bytes per launch that crossed the L2's memory side for this (kind, shape), from the committed rocprofv3 counter
    passes (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_probe.py + tools/pmc_summary.py; the
    counters cannot be read from inside this process).  Mean over the epilogue variants measured; None when this
    shape / batch / dtype was not part of the counter run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
    if not os.path.exists(path):
        return None
    tab = json.load(open(path))
    if tab.get('dtype') != dtype:
        return None
    s = entry.shape
    key = dict(n=s.n, d=s.d, h=s.h, w=s.w, cin=s.cin, cout=s.cout, k=[s.kd, s.kh, s.kw])
    kind = 'fwd' if entry.kind == 0 else 'wgrad'
    hits = [e['traffic_bytes'] for e in tab['entries'] if e['kind'] == kind and e['shape'] == key]
    if not hits or s.upsample_in:
        return None
    return round(sum(hits) / len(hits))


def main():
    args = parse()
    from saragan_amd import _lib, parallel
    rank, world, local = parallel.init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))   # (rehearsals may stack ranks on one GPU)
    torch.cuda.set_device(device)
    cfg = build(args, device)
    sess, ph = cfg['sess'], cfg['ph']
    batches = [synthetic_batch(cfg['shape'], rank * 1000 + i, device) for i in range(4)]

    def step(i):
        sess.run(cfg['train'], feed_dict={ph: batches[i % len(batches)]})
        sess.run(cfg['ema_op'])

    for i in range(args.warmup):
        step(i)
    lib = _lib.load()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    lib.sg_prof_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    # kernel timings (HIP events recorded on the launch stream inside the timed region)
    import ctypes as C
    ents = (_lib.ProfEntry * 256)()
    n_ent = C.c_int32(0)
    lib.sg_prof_collect(ents, 256, C.byref(n_ent))
    lib.sg_prof_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    if args.dump_prof:
        rows = sorted((ents[i] for i in range(n_ent.value)), key=lambda e: -e.total_ms)
        for e in rows:
            s_ = e.shape
            print(f"{'fwd ' if e.kind == 0 else 'wgrd'} n{s_.n} {s_.d}x{s_.h}x{s_.w} {s_.cin:4d}->{s_.cout:4d} k{s_.kd}{s_.kh}{s_.kw} "
                  f"ups{s_.upsample_in} calls/step {e.launches / args.steps:5.1f} avg {e.total_ms / e.launches * 1e3:8.1f} us "
                  f"ms/step {e.total_ms / args.steps:7.3f} TF/s {e.flops_per_launch / (e.total_ms / e.launches) / 1e9:7.1f}",
                  file=sys.stderr)
    vols = args.batch * world * args.steps
    value = vols / dt
    # dominant kernel = the (kind, shape) with the largest total time
    best = None
    for i in range(n_ent.value):
        e = ents[i]
        if best is None or e.total_ms > best.total_ms:
            best = e
    peak = 2500.0 if args.dtype == 'bf16' else 157.3
    roof = None
    if best is not None and best.launches > 0:
        avg_ms = best.total_ms / best.launches
        ach = best.flops_per_launch / (avg_ms * 1e-3) / 1e12
        s = best.shape
        roof = dict(bound='mfma', achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4),
                    traffic=pmc_traffic(best, args.dtype),
                    kernel=('conv_fwd_kernel' if best.kind == 0 else 'conv_wgrad_kernel'),
                    shape=dict(n=s.n, d=s.d, h=s.h, w=s.w, cin=s.cin, cout=s.cout, k=[s.kd, s.kh, s.kw],
                               upsample_in=s.upsample_in),
                    launches=int(best.launches), avg_ms=round(avg_ms, 4),
                    flops_per_launch=best.flops_per_launch)
    fg, fd = conv_flops_per_volume(cfg['ks'], cfg['fs'], args.phase, cfg['base_shape'], args.latent)
    # executed conv work: G fwd+dgrad+wgrad; D: 3 forwards, 3 (wgan: the G loss reuses the D-loss data gradient)
    # or 4 data-gradient passes, 2 weight-gradient passes, 2 convs of the GP double backward + its weight gradient
    step_gf = (3 * fg + (11 if args.loss == 'wgan' else 12) * fd) / 1e9
    total_conv_ms = sum(ents[i].total_ms for i in range(n_ent.value))
    out = dict(metric='3D volumes/sec/node (G+D step) at 128x128x32', value=round(value, 3), unit='volumes/s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
               higher_is_better=True, scaling='weak', vs_baseline=None, dtype=args.dtype, data='synthetic',
               config=dict(workload=f"pgan '{args.size}' phase {args.phase} G+D step, volumes "
                                    f"{cfg['shape'][2]}x{cfg['shape'][3]}x{cfg['shape'][4]}, {args.loss}-gp, "
                                    f"simultaneous, alpha {args.alpha}",
                           local_batch=args.batch, global_batch=args.batch * world, latent_dim=args.latent,
                           parallelism=f'dp{world}', step_gflop_per_volume=round(step_gf, 1),
                           step_mfma_tflops=round(value * step_gf / 1e3 / world, 2),
                           conv_kernel_ms_per_step=round(total_conv_ms / args.steps, 3)),
               roofline=roof)
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args, cfg, args.cpu_budget_s)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
