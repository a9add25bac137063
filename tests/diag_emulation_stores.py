"""Diagnostic (test infrastructure, not collected): the tensors the bf16 HIP path STORES during one generator / discriminator
forward, in order, against the tensors the oracle's bf16 emulation rounds (`_q`), in order.  The two lists are aligned by shape; the
first pair that differs by more than rare 1-ulp flips names the rounding point the emulation misses.

    python tests/diag_emulation_stores.py <phase> [G|D]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pgan_oracle as O  # noqa: E402

BASE = (1, 1, 4, 4)
LATENT = 512


def main():
    from saragan_amd import functional as F
    from saragan_amd.networks import ops
    from saragan_amd.networks.ops import ScalarVariable, materialize
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    phase = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    which = sys.argv[2] if len(sys.argv) > 2 else 'G'
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ks, fs = O.preset_specs('s', BASE, 8)
    p32 = O.init_params(phase, BASE, LATENT, ks, fs, seed=100 + phase, dtype=torch.float32)
    g = torch.Generator().manual_seed(phase)
    z = torch.randn(2, LATENT, generator=g)
    img = (BASE[0], *[d * 2 ** (phase - 1) for d in BASE[1:]])
    x_in = torch.randn(2, *img, generator=g)
    # ---- emulation: every _q in order
    emu_log = []
    q0 = O._q

    def q_rec(x):
        y = q0(x)
        emu_log.append(y.detach().float().clone())
        return y
    O._q = q_rec
    with O.bf16_emulation(), torch.no_grad():
        if which == 'G':
            ye = O.generator(p32, z, 0.0, phase, BASE, 'leaky_relu', ks, fs, 0.2)
        else:
            ye = O.discriminator(p32, O._q(x_in), 0.0, phase, LATENT, 'leaky_relu', ks, fs, 0.2)
    O._q = q0
    # ---- HIP: every tensor an op returns, in order
    hip_log = []

    def wrap(mod, name):
        f0 = getattr(mod, name)

        def f(*a, **k):
            out = f0(*a, **k)
            for t in (out if isinstance(out, tuple) else (out,)):
                if torch.is_tensor(t):
                    hip_log.append((name, t.detach().float().cpu()))
            return out
        setattr(mod, name, f)
    for name in ('conv3d', 'pixel_norm', 'conv3d_pn_to_rgb', 'conv3d_act_pool', 'dense', 'bias_act', 'downscale2x', 'upscale2x'):
        if hasattr(F, name):
            wrap(F, name)
    set_compute_dtype(torch.bfloat16)
    store = VariableStore('cuda', seed=0)
    alpha = ScalarVariable(0.0, 'alpha')
    with use_store(store), torch.no_grad():
        if which == 'G':
            materialize(generator(z.cuda(), alpha, phase, BASE, 'leaky_relu', ks, fs, 0.2))
        else:
            materialize(discriminator(x_in.cuda(), alpha, phase, LATENT, 'leaky_relu', ks, fs, 0.2))
    store.load_state_dict({k: v for k, v in p32.items()}, strict=True)
    F.clear_pack_cache()
    hip_log.clear()
    with use_store(store), torch.no_grad():
        if which == 'G':
            yh = materialize(generator(z.cuda(), alpha, phase, BASE, 'leaky_relu', ks, fs, 0.2, is_reuse=True))
        else:
            yh = materialize(discriminator(x_in.cuda(), alpha, phase, LATENT, 'leaky_relu', ks, fs, 0.2, is_reuse=True))
    print('emulation rounds', len(emu_log), 'tensors; HIP ops returned', len(hip_log))
    for i, t in enumerate(emu_log):
        print('  emu', i, tuple(t.shape))
    for i, (n, t) in enumerate(hip_log):
        print('  hip', i, n, tuple(t.shape))
    # align: walk the HIP list, match each to the next emulation tensor with the same element count
    j = 0
    for i, (n, t) in enumerate(hip_log):
        k = j
        while k < len(emu_log) and emu_log[k].numel() != t.numel():
            k += 1
        if k == len(emu_log):
            print(f'hip {i} {n} {tuple(t.shape)}: no emulation tensor left with this size')
            continue
        e = emu_log[k].reshape(t.shape) if emu_log[k].shape != t.shape else emu_log[k]
        d = float((t.double() - e.double()).norm() / e.double().norm().clamp_min(1e-30))
        ne = float((t != e).float().mean())
        print(f'hip {i} {n:18s} {str(tuple(t.shape)):28s} <-> emu {k}: rel L2 {d:.2e}, entries differing {ne:.4f}')
        j = k + 1
    print('output: rel', float((yh.float().cpu().double() - ye.double()).norm() / ye.double().norm()))


if __name__ == '__main__':
    main()
