"""Gradient penalty of the benchmarked discriminator (pgan 's' phase 6, bf16, 32x128x128) with and without the fused masked
gather in its first backward: the penalty and every parameter gradient of the double backward side by side.  Diagnostic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import functional as F   # noqa: E402
from saragan_amd.networks.pgan.discriminator import discriminator   # noqa: E402
from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store   # noqa: E402
from saragan_amd.networks.pgan.variables import preset_specs   # noqa: E402


def run(no_gather, n=4):
    F._NO_GATHER_BWD = no_gather
    set_compute_dtype(torch.bfloat16)
    store = VariableStore('cuda', seed=3)
    torch.manual_seed(5)
    x = torch.randn(n, 1, 32, 128, 128, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last_3d)
    with use_store(store):
        xi = x.clone().requires_grad_(True)
        ks, fs = preset_specs('s', (1, 1, 4, 4), 8)
        d = discriminator(xi, 0.0, 6, 512, 'leaky_relu', ks, fs, param=0.2).float()
        params = [v for v in store.vars.values()]
        with F.skip_param_grads(params):
            (gr,) = torch.autograd.grad(d, xi, grad_outputs=torch.ones_like(d), create_graph=True)
        slopes = torch.sqrt(F.sumsq_keep_w(gr).sum(dim=1))
        gp = 10 * ((slopes - 1) ** 2).mean()
        grads = torch.autograd.grad(gp, params, allow_unused=True)
    return float(gp), {k: (g.detach().float().clone() if g is not None else None) for k, g in zip(store.vars.keys(), grads)}, gr.detach().float().clone()


def main():
    gp0, g0, gr0 = run(True)
    gp1, g1, gr1 = run(False)
    print('gp', gp0, gp1, 'first-backward gradient equal:', bool(torch.equal(gr0, gr1)),
          float((gr0 - gr1).abs().max() / gr0.abs().max()))
    worst = 0.0
    for k in g0:
        if g0[k] is None or g1[k] is None:
            print(k, 'None', g0[k] is None, g1[k] is None)
            continue
        e = float(torch.linalg.vector_norm(g0[k] - g1[k]) / (torch.linalg.vector_norm(g0[k]) + 1e-30))
        worst = max(worst, e)
        if e > 1e-4:
            print(f'{k}: rel L2 {e:.3e}')
    print('worst rel L2', worst)


if __name__ == '__main__':
    main()
