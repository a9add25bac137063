"""Diagnostic (test infrastructure, not collected): where does the bf16 HIP path leave the oracle's bf16 emulation?

For phases 1..P of pgan 's' (the benchmarked network's filters) at batch 2: the generator's output image and the
discriminator's logits + its input gradient, on the HIP path (bf16) against oracle.bf16_emulation() and the fp64 oracle, with the
SAME weights and inputs.  A faithful emulation differs from the HIP result by rare 1-ulp flips (relative L2 far below its own
distance from fp64); the first phase where `hip-emu` comes close to `emu-f64` names the block whose rounding points differ.

    python tests/diag_emulation_layers.py [max_phase]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pgan_oracle as O  # noqa: E402

BASE = (1, 1, 4, 4)
LATENT = 512


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    from saragan_amd import functional as F
    from saragan_amd.networks.ops import ScalarVariable, materialize
    from saragan_amd.networks.pgan.discriminator import discriminator
    from saragan_amd.networks.pgan.generator import generator
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    maxp = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ks, fs = O.preset_specs('s', BASE, 8)
    for phase in range(1, maxp + 1):
        p32 = O.init_params(phase, BASE, LATENT, ks, fs, seed=100 + phase, dtype=torch.float32)
        p64 = {k: v.double() for k, v in p32.items()}
        g = torch.Generator().manual_seed(phase)
        z = torch.randn(2, LATENT, generator=g)
        img = (BASE[0], *[d * 2 ** (phase - 1) for d in BASE[1:]])
        x_in = torch.randn(2, *img, generator=g)
        # oracle: fp64 and the emulation
        y64 = O.generator(p64, z.double(), 0.0, phase, BASE, 'leaky_relu', ks, fs, 0.2)
        with O.bf16_emulation():
            ye = O.generator(p32, z, 0.0, phase, BASE, 'leaky_relu', ks, fs, 0.2)

        def d_oracle(p, x, emu):
            x = x.clone().requires_grad_(True)
            ctx = O.bf16_emulation() if emu else torch.enable_grad()
            with ctx:
                xin = O._q(x) if emu else x
                logit = O.discriminator(p, xin, 0.0, phase, LATENT, 'leaky_relu', ks, fs, 0.2)
                (gx,) = torch.autograd.grad(logit.sum(), x)
            return logit.detach(), gx
        l64, gx64 = d_oracle(p64, x_in.double(), False)
        le, gxe = d_oracle(p32, x_in, True)
        # HIP, bf16
        set_compute_dtype(torch.bfloat16)
        store = VariableStore('cuda', seed=0)
        alpha = ScalarVariable(0.0, 'alpha')
        with use_store(store):
            yh = materialize(generator(z.cuda(), alpha, phase, BASE, 'leaky_relu', ks, fs, 0.2))
            xh = x_in.cuda().requires_grad_(True)
            lh = materialize(discriminator(xh, alpha, phase, LATENT, 'leaky_relu', ks, fs, 0.2))
        store.load_state_dict({k: v for k, v in p32.items()}, strict=True)
        F.clear_pack_cache()
        with use_store(store):      # (again, with the loaded weights)
            yh = materialize(generator(z.cuda(), alpha, phase, BASE, 'leaky_relu', ks, fs, 0.2, is_reuse=True)).float().cpu()
            lh_t = materialize(discriminator(xh, alpha, phase, LATENT, 'leaky_relu', ks, fs, 0.2, is_reuse=True))
            (gxh,) = torch.autograd.grad(lh_t.float().sum(), xh)
        lh, gxh = lh_t.detach().float().cpu(), gxh.float().cpu()
        set_compute_dtype(torch.float32)
        ne = float((yh != ye).float().mean())
        print(f'phase {phase} {img}:  G out  hip-emu {rel(yh, ye):.2e} (entries differing {ne:.3f})  emu-f64 {rel(ye, y64):.2e}  hip-f64 {rel(yh, y64):.2e}')
        print(f'            D logit hip-emu {rel(lh, le):.2e}  emu-f64 {rel(le, l64):.2e}  hip-f64 {rel(lh, l64):.2e}   '
              f'D input gradient hip-emu {rel(gxh, gxe):.2e}  emu-f64 {rel(gxe, gx64):.2e}  hip-f64 {rel(gxh, gx64):.2e}', flush=True)
        del store
        F.clear_pack_cache()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
