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

This file holds the step and the timed region; the rest is in benchlib/: launch.py (arguments, rank spawning, --dry-run),
workload.py (networks, optimizers, synthetic batches, the state the legs start from), legs.py (the timed region's bracket, the
settle and calibration steps, the extra legs, the CPU baseline -- the only place outside tests/ and smoke() that imports oracle/),
roofline.py (the dominant kernel's roofline object).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import legs, roofline                                          # noqa: E402
from benchlib.launch import CONFIGS, dry_run, parse, spawn_ranks             # noqa: E402,F401
from benchlib.workload import (build, conv_flops_per_volume, restore_state,  # noqa: E402,F401
                               snapshot_state, synthetic_batch, synthetic_volume)

# The host enqueues a step in ~10 ms and the device runs it in ~58: unthrottled, a long run has the host hundreds of steps ahead and
# the caching allocator cannot hand blocks back that are still queued (700 steps ended in an out-of-memory error at 278 GiB).  The
# host waits for the step issued RUN_AHEAD steps earlier: the queue never drains, so nothing changes for the device.
RUN_AHEAD = 6
NCAL = 2      # eager calibration steps before the timed region (legs.calibrate)


def main():
    args = parse()
    if args.dry_run:
        os.environ['SARAGAN_DIST_BACKEND'] = 'gloo'
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args, os.path.abspath(__file__)))
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
    batches = [synthetic_batch(cfg['shape'], rank * 1000 + i, device) for i in range(4)]      # resident in HBM
    lib = _lib.load()
    inflight = []

    def step(i):
        """One pass of the hot path over one resident batch: G + D step (simultaneous), then the EMA of the generator."""
        sess.run(cfg['train'], feed_dict={ph: batches[i % len(batches)]})
        sess.run(cfg['ema_op'])
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > RUN_AHEAD:
            inflight.pop(0).synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- W untimed warm-up steps; settle steps for the board (the model state is put back); two eager calibration steps
    for i in range(args.warmup):
        step(i)
    snap = snapshot_state(cfg)      # the state the timed region starts from (and every extra leg)
    preheat = legs.settle(step, args.warmup, world)
    if preheat['steps']:
        # the settle steps were for the board, not for the model: the timed region trains on from the state after the W warm-up
        # steps (this is WGAN-GP at lr 1e-3 on noise: every extra step takes the trajectory further towards the edge of the finite range)
        restore_state(cfg, snap)
    table = legs.calibrate(lib, step, args.warmup, NCAL, barrier)
    hipgraph = any('graph' in e for e in cfg['train'][0].graph.__dict__.get('_captures', {}).values())
    dom_name, dom, by_kernel = roofline.dominant(table)
    if dom is not None:      # the timed region brackets only the dominant (kind, shape)'s launches
        lib.sg_prof_set_filter(dom.kind, C.byref(dom.shape))
    barrier()
    if dp:
        for o_ in cfg['optimizers']:
            o_.distributed.exposed_ms()       # forget the warm-up steps
    # (a captured step is replayed as one launch: nothing to bracket, the roofline object then quotes the calibration steps)
    lib.sg_prof_enable(0 if (os.environ.get('SARAGAN_BENCH_NO_PROF') or hipgraph) else 1)

    # ---- THE TIMED REGION: exactly K steps between barrier + synchronize on both sides (the barrier before is the one above)
    with legs.quiet_collector(), legs.Stopwatch(lambda: None) as sw:
        for i in range(args.steps):
            step(args.warmup + NCAL + i)
        barrier()
    dt = sw.seconds      # the longer of wall clock and HIP events (legs.Stopwatch)

    if dp:      # all-reduce time left exposed behind backward, per step (G + D), this rank
        comm['exposed_allreduce_ms_per_step'] = round(sum(sum(o_.distributed.exposed_ms()) for o_ in cfg['optimizers']) / args.steps, 3)
        comm['bucket_mib'] = cfg['optimizers'][0].distributed.bucket_elems * 4 >> 20
    timed = roofline.collect(lib)
    lib.sg_prof_enable(0)
    lib.sg_prof_set_filter(0, None)
    losses = [float(v) for v in sess.run(cfg['losses'] + cfg['train'], feed_dict={ph: batches[0]})[:2]]
    if not all(l == l and abs(l) < 1e30 for l in losses):
        raise SystemExit(f'non-finite losses after the timed steps: {losses}')
    if dp:      # MAX over ranks
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return

    # ---- the line
    if args.dump_prof:
        roofline.dump_table(table, NCAL)
    value = args.batch * world * args.steps / dt
    fg, fd = conv_flops_per_volume(cfg['ks'], cfg['fs'], args.phase, cfg['base_shape'], args.latent, args.dims)
    # executed conv work: G fwd+dgrad+wgrad; D: 3 forwards, 3 (wgan: the G loss reuses the D-loss data gradient)
    # or 4 data-gradient passes, 2 weight-gradient passes, 2 convs of the GP double backward + its weight gradient
    step_gf = (3 * fg + (11 if args.loss == 'wgan' else 12) * fd) / 1e9
    sh = cfg['shape']
    dims = 'x'.join(str(v) for v in (sh[3], sh[4], sh[2])) if args.dims == 3 else f'{sh[3]}x{sh[4]}'
    out = dict(metric=f"3D volumes/sec/node (G+D step) at {dims}" if args.dims == 3 else f"2D images/sec/node (G+D step) at {dims}",
               value=round(value, 3), unit='volumes/s' if args.dims == 3 else 'images/s',
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
                           conv_kernel_ms_per_step=round(sum(e.total_ms for e in table) / NCAL, 3),
                           conv_ms_per_step_by_kernel={k.decode(): round(v / NCAL, 3) for k, v in
                                                       sorted(by_kernel.items(), key=lambda kv: -kv[1])[:8]},
                           losses_after=dict(disc=round(losses[0], 4), gen=round(losses[1], 4))),
               roofline=roofline.roofline_object(timed, table, dom_name, args.dtype, hipgraph, NCAL))
    small = roofline.small_channel_object(table, args.dtype)
    if small:
        out['roofline_hbm'] = small
    if world == 1 and args.config == 3 and not args.no_extras:
        cpu_cfg = dict(ks=cfg['ks'], fs=cfg['fs'], base_shape=cfg['base_shape'], shape=cfg['shape'])      # (what cpu_baseline needs, kept past the legs)
        extras = {}
        try:
            restore_state(cfg, snap)      # (every leg starts where the main loop did: after the warm-up)
            extras['loader_in_loop'] = legs.loader_leg(args, cfg, device, max(3, args.steps), barrier, snap)
            if float(args.alpha) in (0.0, 1.0):
                extras['fade_branch_computed'] = legs.fade_leg(args, cfg, snap, step, batches[0], barrier)
            del cfg, sess, batches, step
            from saragan_amd import functional as F
            F.clear_pack_cache()
            torch.cuda.empty_cache()
            extras['f32'] = legs.f32_leg(args, device, lib, barrier, step_gf, NCAL)
        except Exception as exc:      # an extra leg must never cost the headline line
            import traceback
            traceback.print_exc()
            extras['error'] = repr(exc)[:300]
        out['extras'] = extras
        cfg = cpu_cfg
    if world == 1 and not args.no_cpu_baseline:
        try:
            out['cpu_baseline'] = legs.cpu_baseline(args, cfg, args.cpu_budget_s)
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
