"""bench.py: the workload -- the networks, optimizers and step of one BASELINE configuration, synthetic LIDC-shaped batches,
the FLOP count of a step, and the state every leg starts from."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
