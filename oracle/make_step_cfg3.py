"""Generates tests/golden/oracle_step_cfg3_n2.npz: ONE whole `simultaneous` optimisation step (reference
optimization.py:128-163, networks/loss.py:101-165) of the BENCHMARKED network -- BASELINE configs[2]: pgan 's' phase 6, volumes
32 x 128 x 128, latent 512, WGAN-GP 10, alpha 0, Adam(0, 0.9) lr 1e-3 -- at batch 2 on the CPU oracle, in fp64 and in the
bf16-EMULATING arithmetic (fp32 with the HIP path's rounding points, pgan_oracle.bf16_emulation).

TEST INFRASTRUCTURE ONLY (nothing under saragan_amd/ imports this).  A whole gradient set of this network is 2 x 28.9 M numbers;
the fixture keeps, per variable, what pins it without shipping it:
  * the L2 norm and the plain sum of its gradient,
  * SAMPLES entries of the gradient at fixed pseudo-random positions (seeded by the variable's name: `sample_index`),
  * the post-Adam weight and the EMA shadow at the same positions,
plus the three losses, and for gen_sample its sum, sum of squares and SAMPLES voxels.  (< 2 MB compressed.)
The inputs are not stored: tests regenerate them from the seeds (oracle/make_loss_curve.py: cfg3_setup / cfg3_inputs).

An fp64 step of this size takes a few minutes on 8 cores and ~25 GB.   usage: python oracle/make_step_cfg3.py [f64] [bf16emu] [bf16emu64]
       python oracle/make_step_cfg3.py cfg4 [f32] [bf16emu]      (tests/golden/oracle_step_cfg4_n1.npz: configs[3] at batch 1)
"""
import contextlib
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_loss_curve as MC  # noqa: E402
from oracle import pgan_oracle as O  # noqa: E402

SAMPLES = 1024
OUT = os.path.join(ROOT, 'tests', 'golden', 'oracle_step_cfg3_n2.npz')
OUT4 = os.path.join(ROOT, 'tests', 'golden', 'oracle_step_cfg4_n1.npz')


def cfg4_setup(dtype=torch.float32):
    """BASELINE configs[3] at batch 1: pgan 'm' phase 7 (volumes 64 x 256 x 256, 101.7 M parameters), latent 512, WGAN-GP 10,
    MIXING (alpha 0.5: both fade-in branches run) with the freeze train ops (only the new phase's variables are updated, Q4).
    An fp64 step of this size does not fit the build container (62 GB, no swap: the fp32 step peaks at 25.7 GB, an fp64 one would need twice that), so the two
    arithmetics of this fixture are fp32 and the bf16 emulation (fp32 sums)."""
    base = MC.BASE
    ks, fs = O.preset_specs('m', base, 8)
    phase, latent, n, seed = 7, 512, 1, 4107
    p0 = O.init_params(phase, base, latent, ks, fs, seed=seed, dtype=dtype, bias_std=0.05)
    img = (base[0], *[d * 2 ** (phase - 1) for d in base[1:]])
    rnd = O.draw_randomness(n, latent, img, seed + 1, dtype=dtype)
    rng = np.random.default_rng(1234 + seed)
    vol = np.clip(rng.normal(1024, 512, (n, *img)), 0, 4095).astype(np.int16).astype(np.float64)
    real = torch.as_tensor((vol - 1024.0) / 1024.0).to(dtype)
    cfg = dict(phase=phase, base_shape=base, latent_dim=latent, kernel_spec=ks, filter_spec=fs, activation='leaky_relu',
               leakiness=0.2, loss_fn='wgan', gp_weight=10.0, noise_stddev=0.01)
    freeze = list(O.variable_shapes(phase - 1, base, latent, ks, fs).keys())
    return dict(p0=p0, rnd=rnd, real=real, alpha=0.5, cfg=cfg, freeze=freeze, phase=phase, n=n, latent=latent, img=img,
                kernel_spec=ks, filter_spec=fs, lr=1e-3)


def run4(arith):
    s = cfg4_setup(torch.float32)
    p = {k: v.clone() for k, v in s['p0'].items()}
    shadow = {k: v.clone() for k, v in s['p0'].items()}
    ag, ad = O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9)
    emu = O.bf16_emulation() if arith == 'bf16emu' else contextlib.nullcontext()
    t0 = time.time()
    with emu:
        out = O.step_simultaneous(p, ag, ad, shadow, s['rnd'], s['real'], s['alpha'], s['cfg'], s['lr'], s['lr'], freeze=s['freeze'])
    import resource
    print('cfg4', arith, 'step took', round(time.time() - t0, 1), 's, peak RSS',
          round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1), 'GB', flush=True)
    res = {f'{arith}:gen_loss': float(out['gen_loss']), f'{arith}:disc_loss': float(out['disc_loss']),
           f'{arith}:gp_loss': out['gp_loss'].double().reshape(-1).numpy()}
    gs = out['gen_sample'].double().reshape(-1)
    res[f'{arith}:gen_sample_sum'] = float(gs.sum())
    res[f'{arith}:gen_sample_sumsq'] = float((gs * gs).sum())
    res[f'{arith}:gen_sample_at'] = gs[torch.as_tensor(sample_index('gen_sample', gs.numel()))].numpy().astype(np.float32)
    grads = dict(out['g_grads'])
    grads.update(out['d_grads'])
    for k, g in grads.items():      # (the trained variables only: the previous phase's are frozen)
        gd = g.double().reshape(-1)
        idx = torch.as_tensor(sample_index(k, gd.numel()))
        res[f'{arith}:gnorm:{k}'] = float(gd.norm())
        res[f'{arith}:gsum:{k}'] = float(gd.sum())
        res[f'{arith}:g:{k}'] = gd[idx].numpy().astype(np.float32)
        res[f'{arith}:w:{k}'] = p[k].double().reshape(-1)[idx].numpy().astype(np.float32)
        res[f'{arith}:ema:{k}'] = shadow[k].double().reshape(-1)[idx].numpy().astype(np.float32)
    return res


def sample_index(name, numel):
    """Positions (into the flattened tensor, reference layout) of the kept entries of variable `name`."""
    if numel <= SAMPLES:
        return np.arange(numel, dtype=np.int64)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, SAMPLES, replace=False)).astype(np.int64)


def run(arith):
    # 'bf16emu64': the emulation's rounding points with fp64 accumulation -- not a reference: its distance from 'bf16emu' (the same
    # rounding points, f32 accumulation) is what the ORDER / PRECISION of the sums alone is worth after the network's depth
    dtype = torch.float64 if arith in ('f64', 'bf16emu64') else torch.float32
    s = MC.cfg3_setup(dtype, 'cfg3')
    p = {k: v.clone() for k, v in s['p0'].items()}
    shadow = {k: v.clone() for k, v in s['p0'].items()}
    ag, ad = O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9)
    real, rnd = MC.cfg3_inputs(s, 0, dtype)
    emu = O.bf16_emulation() if arith in ('bf16emu', 'bf16emu64') else contextlib.nullcontext()
    t0 = time.time()
    with emu:
        out = O.step_simultaneous(p, ag, ad, shadow, rnd, real, s['alpha'], s['cfg'], s['lr'], s['lr'])
    print(arith, 'step took', round(time.time() - t0, 1), 's', flush=True)
    res = {f'{arith}:gen_loss': float(out['gen_loss']), f'{arith}:disc_loss': float(out['disc_loss']),
           f'{arith}:gp_loss': out['gp_loss'].double().reshape(-1).numpy()}
    gs = out['gen_sample'].double().reshape(-1)
    res[f'{arith}:gen_sample_sum'] = float(gs.sum())
    res[f'{arith}:gen_sample_sumsq'] = float((gs * gs).sum())
    res[f'{arith}:gen_sample_at'] = gs[torch.as_tensor(sample_index('gen_sample', gs.numel()))].numpy().astype(np.float32)
    grads = dict(out['g_grads'])
    grads.update(out['d_grads'])
    for k, g in grads.items():
        gd = g.double().reshape(-1)
        idx = torch.as_tensor(sample_index(k, gd.numel()))
        res[f'{arith}:gnorm:{k}'] = float(gd.norm())
        res[f'{arith}:gsum:{k}'] = float(gd.sum())
        res[f'{arith}:g:{k}'] = gd[idx].numpy().astype(np.float32)
        res[f'{arith}:w:{k}'] = p[k].double().reshape(-1)[idx].numpy().astype(np.float32)
        res[f'{arith}:ema:{k}'] = shadow[k].double().reshape(-1)[idx].numpy().astype(np.float32)
    return res


if __name__ == '__main__':
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', '8')))
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg4':
        have = dict(np.load(OUT4)) if os.path.exists(OUT4) else {}
        for arith in (sys.argv[2:] or ['f32', 'bf16emu']):
            have = {k: v for k, v in have.items() if not k.startswith(arith + ':')}
            have.update(run4(arith))
            np.savez_compressed(OUT4, **have)
        print(OUT4, os.path.getsize(OUT4), 'bytes')
        sys.exit(0)
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for arith in (sys.argv[1:] or ['f64', 'bf16emu', 'bf16emu64']):
        have = {k: v for k, v in have.items() if not k.startswith(arith + ':')}
        have.update(run(arith))
        np.savez_compressed(OUT, **have)
    print(OUT, os.path.getsize(OUT), 'bytes')
