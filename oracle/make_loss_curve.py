"""Generates tests/golden/loss_curve_*.npz: G and D loss curves of the CPU oracle over 200+ optimisation steps of the
toy pgan of tests/stepfix.py ("bf16 path compared against fp32 oracle ... with loss-curve
agreement over >= 200 steps"; BASELINE north_star: "G+D loss curves matching the CPU reference within tolerance").

TEST INFRASTRUCTURE ONLY.  Each curve is produced twice, in fp64 and in fp32 (the reference's arithmetic, ops.py:147-150):
the distance between the two is the divergence that rounding alone causes on this trajectory, and the GPU test states
its tolerance against it.  All randomness is injected per step (loss.py:116-133 cannot be matched across RNGs):
`curve_inputs(step)` below is what the GPU test feeds the HIP path.

usage: python oracle/make_loss_curve.py            (writes tests/golden/loss_curve_{wgan,logistic_mix}.npz)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pgan_oracle as O  # noqa: E402

BASE = (1, 1, 4, 4)
LATENT = 16
FILTER_SPEC = [[16, 16], [16, 8], [8, 8]]
KERNEL_SPEC = [[[1, 3, 3], [1, 3, 3]], [[1, 3, 3], [3, 3, 3]], [[3, 3, 3], [3, 3, 3]]]
PHASE, N, STEPS, NVOL = 3, 4, 240, 16
CURVES = {
    'wgan': dict(loss_fn='wgan', gp_weight=10.0, alpha=0.0, lr=1e-3),
    'logistic_mix': dict(loss_fn='logistic', gp_weight=1.0, alpha=0.25, lr=1e-3),   # mixing: freeze train ops (Q4)
}


def curve_setup(name, dtype=torch.float64):
    c = CURVES[name]
    img = (BASE[0], *[d * 2 ** (PHASE - 1) for d in BASE[1:]])
    cfg = dict(phase=PHASE, base_shape=BASE, latent_dim=LATENT, kernel_spec=KERNEL_SPEC, filter_spec=FILTER_SPEC,
               activation='leaky_relu', leakiness=0.2, loss_fn=c['loss_fn'], gp_weight=c['gp_weight'], noise_stddev=0.01)
    p0 = O.init_params(PHASE, BASE, LATENT, KERNEL_SPEC, FILTER_SPEC, seed=77, dtype=dtype)
    rng = np.random.default_rng(4321)
    vols = np.clip(rng.normal(1024, 512, (NVOL, *img)), 0, 4095).astype(np.int16).astype(np.float64)
    vols = torch.as_tensor((vols - 1024.0) / 1024.0).to(dtype)
    freeze = None
    if c['alpha'] > 0:
        freeze = list(O.variable_shapes(PHASE - 1, BASE, LATENT, KERNEL_SPEC, FILTER_SPEC).keys())
    return dict(cfg=cfg, p0=p0, vols=vols, freeze=freeze, img=img, **c)


def curve_inputs(setup, step, dtype=torch.float64):
    """(real batch, injected randomness) of optimisation step `step`."""
    idx = [(step * N + i) % NVOL for i in range(N)]
    rnd = O.draw_randomness(N, LATENT, setup['img'], 5000 + step, dtype=dtype)
    return setup['vols'][idx].to(dtype), rnd


def run_oracle(name, dtype):
    s = curve_setup(name, dtype)
    p = {k: v.clone() for k, v in s['p0'].items()}
    ag, ad = O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9)
    g, d = [], []
    for step in range(STEPS):
        real, rnd = curve_inputs(s, step, dtype)
        out = O.step_simultaneous(p, ag, ad, None, rnd, real, s['alpha'], s['cfg'], s['lr'], s['lr'], freeze=s['freeze'])
        g.append(float(out['gen_loss']))
        d.append(float(out['disc_loss']))
    return np.asarray(g), np.asarray(d)


def smooth(x, win=20):
    k = np.ones(win) / win
    return np.convolve(np.asarray(x, dtype=np.float64), k, mode='valid')


if __name__ == '__main__':
    torch.set_num_threads(8)
    for name in CURVES:
        g64, d64 = run_oracle(name, torch.float64)
        g32, d32 = run_oracle(name, torch.float32)
        out = os.path.join(ROOT, 'tests', 'golden', f'loss_curve_{name}.npz')
        np.savez_compressed(out, gen_f64=g64, disc_f64=d64, gen_f32=g32, disc_f32=d32)
        print(name, 'gen range', g64.min(), g64.max(), 'disc range', d64.min(), d64.max())
        print('  f32 vs f64: max |d gen|', np.abs(g32 - g64).max(), 'max |d disc|', np.abs(d32 - d64).max(),
              'smoothed:', np.abs(smooth(g32) - smooth(g64)).max(), np.abs(smooth(d32) - smooth(d64)).max())
