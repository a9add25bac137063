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
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for arith in (sys.argv[1:] or ['f64', 'bf16emu', 'bf16emu64']):
        have = {k: v for k, v in have.items() if not k.startswith(arith + ':')}
        have.update(run(arith))
        np.savez_compressed(OUT, **have)
    print(OUT, os.path.getsize(OUT), 'bytes')
