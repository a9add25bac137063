"""Generates tests/golden/ref_*.npz by RUNNING the reference's importable PyTorch modules
(/root/reference/pgan_pytorch/network_dict.py, loss.py) in the build container, and
tests/golden/oracle_*.npz from oracle/pgan_oracle.py (fp64 master).  TEST INFRASTRUCTURE ONLY.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
The reference never travels to the GPU box; only the .npz data files (inputs, weights, expected
outputs) are committed.  Only data is stored, no reference source text.
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
from oracle import pgan_oracle as O  # noqa: E402

REF = '/root/reference/pgan_pytorch'


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print('wrote', name, {k: v.shape for k, v in out.items()})


def reference_goldens():
    if not hasattr(np, 'product'):
        np.product = np.prod  # NumPy 2 shim needed by network_dict.py:217,328
    sys.path.insert(0, REF)
    import network_dict as R
    import loss as RL
    torch.manual_seed(1234)
    dt = torch.float64
    torch.set_default_dtype(dt)
    leak = R.LEAKINESS

    # --- module-level ops ---
    conv = R.EqualizedConv3d(6, 10, 3, 'leaky_relu', padding=1, param=leak)
    x = torch.randn(2, 6, 3, 5, 4)
    npz('ref_eqconv3d.npz', x=x, weight_oidhw=conv.weight, bias=conv.bias, y=conv(x), leak=leak)
    conv133 = R.EqualizedConv3d(5, 7, (1, 3, 3), 'leaky_relu', padding=(0, 1, 1), param=leak)
    x = torch.randn(2, 5, 1, 4, 4)
    npz('ref_eqconv3d_133.npz', x=x, weight_oidhw=conv133.weight, bias=conv133.bias, y=conv133(x), leak=leak)
    lin = R.EqualizedLinear(12, 9, 'leaky_relu', param=leak)
    x = torch.randn(3, 12)
    npz('ref_eqlinear.npz', x=x, weight_oi=lin.weight, bias=lin.bias, y=lin(x), leak=leak)
    x = torch.randn(2, 8, 2, 3, 3)
    x2 = torch.randn(2, 8, 2, 4, 6)
    gb = R.GeneratorBlock(8, 8, 'leaky_relu', leak)
    db = R.DiscriminatorBlock(8, 8, 'leaky_relu', leak)
    npz('ref_simple_ops.npz', x=x, x2=x2, channel_norm=gb.cn(x), upsample=gb.upsampling(x),
        avgpool=db.downsampling(x2), lrelu=R.activation('leaky_relu')(x), leak=leak)
    blk = R.GeneratorBlock(8, 6, 'leaky_relu', leak)
    h = blk.cn(blk.act(blk.conv1(blk.upsampling(x))))   # upsample->conv->bias->act->norm == TF conv_1 stage
    npz('ref_genblock_stage1.npz', x=x, weight_oidhw=blk.conv1.weight, bias=blk.conv1.bias, y=h, leak=leak)

    # --- whole discriminator, phases 1..3, with fade-in, input gradient and gradient penalty ---
    base_shape = (1, 1, 4, 4)
    num_phases, base_dim, latent = 3, 32, 16
    for phase in (1, 2, 3):
        D = R.Discriminator(phase, num_phases, base_dim, latent, base_shape, 'leaky_relu', param=leak)
        D.double()
        shp = (3, 1, 1 * 2 ** (phase - 1), 4 * 2 ** (phase - 1), 4 * 2 ** (phase - 1))
        real = torch.randn(*shp)
        fake = torch.randn(*shp)
        alpha = 0.3 if phase > 1 else 0.0
        xin = real.clone().requires_grad_(True)
        out = D(xin, alpha)
        (gin,) = torch.autograd.grad(out.sum(), xin)
        torch.manual_seed(99 + phase)
        gp = RL.compute_gradient_penalty(D, real, fake, alpha, gradient_penalty_weight=10)
        torch.manual_seed(99 + phase)
        gamma = torch.rand(shp[0], 1, 1, 1, 1)
        dparams = [q for q in D.parameters()]
        gp_grads = torch.autograd.grad(gp, dparams, allow_unused=True)
        arrs = dict(real=real, fake=fake, alpha=alpha, out=out, grad_in=gin, gp=gp, gamma=gamma, leak=leak,
                    phase=phase, num_phases=num_phases, base_dim=base_dim, latent=latent)
        for (name, q), g in zip(D.named_parameters(), gp_grads):
            arrs['p:' + name] = q
            if g is not None:
                arrs['gpgrad:' + name] = g
        npz(f'ref_discriminator_p{phase}.npz', **arrs)

    # --- phase-1 generator (identical op order to the TF graph at phase 1) ---
    G = R.Generator(1, num_phases, base_dim, latent, base_shape, 'leaky_relu', param=leak)
    G.double()
    z = torch.randn(3, latent)
    arrs = dict(z=z, out=G(z, 0.0), leak=leak, base_dim=base_dim, latent=latent)
    for name, q in G.named_parameters():
        arrs['p:' + name] = q
    npz('ref_generator_p1.npz', **arrs)
    # --- generators of phases 2 and 3 with fade-in, plus the input gradient of sum(out) (round 4).  The port's block runs its
    # second stage as conv -> norm -> act (network_dict.py:287-289): the oracle reproduces these with torch_port_order=True
    for phase in (2, 3):
        G = R.Generator(phase, num_phases, base_dim, latent, base_shape, 'leaky_relu', param=leak)
        G.double()
        z = torch.randn(3, latent, requires_grad=True)
        alpha = 0.3
        out = G(z, alpha)
        (gz,) = torch.autograd.grad((out * torch.linspace(0.5, 1.5, out.numel()).reshape(out.shape)).sum(), z)
        arrs = dict(z=z, out=out, grad_z=gz, alpha=alpha, leak=leak, base_dim=base_dim, latent=latent, phase=phase,
                    num_phases=num_phases)
        for name, q in G.named_parameters():
            arrs['p:' + name] = q
        npz(f'ref_generator_p{phase}.npz', **arrs)
    torch.set_default_dtype(torch.float32)


def oracle_goldens():
    """fp64 master fixtures of the full step from the CPU restatement (all randomness stored)."""
    base_shape = (1, 1, 4, 4)
    latent = 16
    filter_spec = [[16, 16], [16, 8], [8, 8]]
    kernel_spec = [[[1, 3, 3], [1, 3, 3]], [[1, 3, 3], [3, 3, 3]], [[3, 3, 3], [3, 3, 3]]]
    for phase, loss_fn, alpha in ((1, 'wgan', 0.0), (2, 'wgan', 0.6), (3, 'logistic', 0.25), (3, 'wgan', 0.0)):
        p = O.init_params(phase, base_shape, latent, kernel_spec, filter_spec, seed=10 + phase, bias_std=0.1)
        n = 4
        img = (1, 2 ** (phase - 1), 4 * 2 ** (phase - 1), 4 * 2 ** (phase - 1))
        rnd = O.draw_randomness(n, latent, img, seed=20 + phase)
        real = torch.randn(n, *img, generator=torch.Generator().manual_seed(30 + phase), dtype=torch.float64)
        cfg = dict(phase=phase, base_shape=base_shape, latent_dim=latent, kernel_spec=kernel_spec,
                   filter_spec=filter_spec, activation='leaky_relu', leakiness=0.2, loss_fn=loss_fn,
                   gp_weight=10.0 if loss_fn == 'wgan' else 1.0, noise_stddev=0.01)
        arrs = {('p0:' + k): v for k, v in p.items()}
        arrs.update({('rnd:' + k): v for k, v in rnd.items()})
        arrs.update(real=real, alpha=alpha, phase=phase, loss_fn=loss_fn, gp_weight=cfg['gp_weight'])
        adam_g, adam_d = O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9)
        shadow = {k: v.clone() for k, v in p.items()}
        freeze = None
        if alpha > 0 and phase > 1:   # mixing: previous-phase variables frozen (Q4)
            freeze = list(O.variable_shapes(phase - 1, base_shape, latent, kernel_spec, filter_spec).keys())
        for s in range(2):
            res = O.step_simultaneous(p, adam_g, adam_d, shadow, rnd, real, alpha, cfg, 1e-3, 1e-3,
                                      freeze=freeze, ema_beta=0.99)
            if s == 0:
                arrs.update(gen_loss=res['gen_loss'], disc_loss=res['disc_loss'], gp_loss=res['gp_loss'],
                            gen_sample=res['gen_sample'])
                arrs.update({('gg:' + k): v for k, v in res['g_grads'].items()})
                arrs.update({('dg:' + k): v for k, v in res['d_grads'].items()})
            arrs.update({(f'p{s + 1}:' + k): v for k, v in p.items()})
            arrs.update({(f'ema{s + 1}:' + k): v for k, v in shadow.items()})
        npz(f'oracle_step_p{phase}_{loss_fn}_a{int(alpha * 100):03d}.npz', **arrs)


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else None      # `reference`: only the fixtures produced by running the reference
    if os.path.isdir(REF) and only in (None, 'reference'):
        reference_goldens()
    if only in (None, 'oracle'):
        oracle_goldens()
