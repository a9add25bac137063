"""Loader for tests/golden/oracle_step_*.npz (written by oracle/make_golden.py)."""
import numpy as np
import torch

from oracle import pgan_oracle as O

BASE_SHAPE = (1, 1, 4, 4)
LATENT = 16
FILTER_SPEC = [[16, 16], [16, 8], [8, 8]]
KERNEL_SPEC = [[[1, 3, 3], [1, 3, 3]], [[1, 3, 3], [3, 3, 3]], [[3, 3, 3], [3, 3, 3]]]


def load_step_fixture(path, dtype=torch.float32):
    z = np.load(path)
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dtype)
    grp = lambda pre: {k[len(pre):]: t(z[k]) for k in z.files if k.startswith(pre)}
    phase, alpha, loss_fn = int(z['phase']), float(z['alpha']), str(z['loss_fn'])
    cfg = dict(phase=phase, base_shape=BASE_SHAPE, latent_dim=LATENT, kernel_spec=KERNEL_SPEC,
               filter_spec=FILTER_SPEC, activation='leaky_relu', leakiness=0.2, loss_fn=loss_fn,
               gp_weight=float(z['gp_weight']), noise_stddev=0.01)
    freeze = None
    if alpha > 0 and phase > 1:
        freeze = list(O.variable_shapes(phase - 1, BASE_SHAPE, LATENT, KERNEL_SPEC, FILTER_SPEC).keys())
    return dict(p0=grp('p0:'), p1=grp('p1:'), p2=grp('p2:'), ema1=grp('ema1:'), ema2=grp('ema2:'),
                rnd=grp('rnd:'), gg=grp('gg:'), dg=grp('dg:'), real=t(z['real']), alpha=alpha, cfg=cfg,
                freeze=freeze, gen_loss=t(z['gen_loss']), disc_loss=t(z['disc_loss']),
                gp_loss=t(z['gp_loss']), gen_sample=t(z['gen_sample']), phase=phase, loss_fn=loss_fn)
