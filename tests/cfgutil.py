"""Shared builders for the BASELINE.json configuration tests: an oracle-side case (initial weights, injected
randomness, a synthetic LIDC-shaped real batch: SURVEY.md section 8d) and the product-side graph for it, driven
through the reference-shaped API (optimization.optimize_step + Session.run)."""
import numpy as np
import torch

from oracle import pgan_oracle as O

BASE = (1, 1, 4, 4)          # start_shape of every 3-D BASELINE config


def make_case(size, phase, latent, n, alpha=0.0, loss_fn='wgan', gp_weight=10.0, seed=0, filter_spec=None,
              kernel_spec=None, base=BASE, bias_std=0.05, dtype=torch.float64):
    """Everything one step needs, as the oracle wants it (fp64 masters)."""
    nph = max(phase, 1)
    if filter_spec is None:
        kernel_spec, filter_spec = O.preset_specs(size, base, nph)
    p0 = O.init_params(phase, base, latent, kernel_spec, filter_spec, seed=seed, dtype=dtype, bias_std=bias_std)
    dims = [d * 2 ** (phase - 1) for d in base[1:]]
    img = (base[0], *dims)
    rnd = O.draw_randomness(n, latent, img, seed + 1, dtype=dtype)
    rng = np.random.default_rng(1234 + seed)
    vol = np.clip(rng.normal(1024, 512, (n, *img)), 0, 4095).astype(np.int16).astype(np.float64)
    real = torch.as_tensor((vol - 1024.0) / 1024.0).to(dtype)        # --data_mean 1024 --data_stddev 1024
    cfg = dict(phase=phase, base_shape=base, latent_dim=latent, kernel_spec=kernel_spec, filter_spec=filter_spec,
               activation='leaky_relu', leakiness=0.2, loss_fn=loss_fn, gp_weight=gp_weight, noise_stddev=0.01)
    freeze = None
    if alpha > 0 and phase > 1:
        freeze = list(O.variable_shapes(phase - 1, base, latent, kernel_spec, filter_spec).keys())
    return dict(p0=p0, rnd=rnd, real=real, alpha=alpha, cfg=cfg, freeze=freeze, phase=phase, loss_fn=loss_fn,
                n=n, latent=latent, base=base, img=img)


def build_product(case, dtype, strategy='simultaneous', clipping=(False, False), lr=(1e-3, 1e-3), beta=(0.0, 0.9),
                  optimizers=None, arch='pgan', ema_decay=0.99):
    """Product-side graph of `case`: returns (store, 20-tuple, placeholder, ema, session, (optimizer_gen, _disc))."""
    import importlib
    import saragan_amd.optimization as opt
    from saragan_amd.ExtendedEMA import ExtendedEMA
    from saragan_amd.networks import loss as L
    from saragan_amd.networks.ops import ScalarVariable
    from saragan_amd.varstore import VariableStore, set_compute_dtype, use_store
    if '.' in arch:      # e.g. 'networks2d.pgan.spec_api': one module exposing both callables
        mod = importlib.import_module(f'saragan_amd.{arch}')
        generator, discriminator = mod.generator, mod.discriminator
    else:
        generator = importlib.import_module(f'saragan_amd.networks.{arch}.generator').generator
        discriminator = importlib.import_module(f'saragan_amd.networks.{arch}.discriminator').discriminator
    set_compute_dtype(dtype)
    store = VariableStore('cuda', seed=0)
    L.set_random_source(L.InjectedRandom({k: v.float() for k, v in case['rnd'].items()}))
    alpha = ScalarVariable(case['alpha'], 'alpha')
    g_lr, d_lr = ScalarVariable(lr[0], 'g_lr'), ScalarVariable(lr[1], 'd_lr')
    if optimizers is None:
        og, od = opt.AdamOptimizer(g_lr, *beta), opt.AdamOptimizer(d_lr, *beta)
    else:
        og, od = optimizers(g_lr, d_lr)
    c = case['cfg']
    ph = opt.Placeholder([case['n'], *case['img']])
    with use_store(store):
        tup = opt.optimize_step(og, od, generator, discriminator, ph, case['latent'], alpha, case['phase'],
                                case['base'], c['kernel_spec'], c['filter_spec'], 'leaky_relu', 0.2, case['loss_fn'],
                                c['gp_weight'], strategy, clipping[0], clipping[1], c['noise_stddev'],
                                None if case['freeze'] is None else list(case['freeze']))
    store.load_state_dict({k: v for k, v in case['p0'].items()}, strict=True)
    ema = ExtendedEMA(list(store.vars.keys()), ema_decay, graph=tup[0].graph)
    return store, tup, ph, ema, opt.Session('cuda'), (og, od)


def pick(tup, mixing):
    """(train_gen, train_disc, g_gradients, g_variables, d_gradients, d_variables, max_g_norm, max_d_norm) of the
    full or the freeze variant of the 20-tuple (optimization.py:221-224)."""
    if mixing:
        return tup[12], tup[16], tup[13], tup[14], tup[17], tup[18], tup[15], tup[19]
    return tup[0], tup[1], tup[6], tup[7], tup[8], tup[9], tup[10], tup[11]


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float(torch.linalg.vector_norm(a - b) / max(float(torch.linalg.vector_norm(b)), 1e-30))


def assert_adam_close(got, ref, lr, rtol, what, max_flip_frac=2e-4):
    """Post-Adam weights: element-wise rtol (+ 2e-5 absolute) against the oracle, except that with beta1 = 0 the first
    updates are -lr * sign(g), so an element whose gradient is zero to rounding may land 2*lr away
    in either arithmetic: at most `max_flip_frac` of a tensor's elements (at least one) may do that, none may be
    further off."""
    g = torch.as_tensor(got).detach().double().cpu().reshape(-1)
    r = torch.as_tensor(ref).detach().double().cpu().reshape(-1)
    diff = (g - r).abs()
    ok = diff <= (2e-5 + rtol * r.abs())
    nbad = int((~ok).sum())
    if nbad:
        assert float(diff.max()) <= 2.2 * lr + 2e-5, (what, float(diff.max()))
        assert nbad <= max(1, int(max_flip_frac * g.numel())), (what, nbad, g.numel())


def bf16_gradient_report(groups, w_rel=0.30, w_cos=0.95, net_cos=0.98, net_norm=0.10):
    """bf16 gradients against the fp64 oracle's.  groups: [(tag, handle_vars, grads, {name: ref})].  bf16 rounding of an
    activation (0.4 %) flips a fraction of a percent of the LeakyReLU masks against the oracle, so single tensors carry
    5-20 % error and bias gradients (sums with cancellation) more.  Criteria: every WEIGHT gradient within `w_rel` in
    relative L2 and cosine >= `w_cos`; each network's whole gradient (all tensors concatenated) cosine >= `net_cos`
    and norm within `net_norm`.  Returns (report, violations)."""
    report, bad = {}, []
    for tag, hv, grads, refs in groups:
        assert [v.key for v in hv] == list(refs.keys())
        a = torch.cat([g.detach().double().cpu().reshape(-1) for g in grads])
        b = torch.cat([refs[v.key].double().reshape(-1) for v in hv])
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        report[f'{tag}:all'] = dict(cos=cos, norm_ratio=float(a.norm() / b.norm()))
        if cos < net_cos or abs(float(a.norm() / b.norm()) - 1) > net_norm:
            bad.append((tag, report[f'{tag}:all']))
        for v, g in zip(hv, grads):
            r = refs[v.key].double()
            gd = g.detach().double().cpu()
            if float(r.norm()) == 0.0:       # e.g. to_rgb_{p-1} at alpha = 0: exactly zero in both
                assert float(gd.norm()) == 0.0, v.key
                continue
            e = rel_l2(gd, r)
            c = float(torch.dot(gd.reshape(-1), r.reshape(-1)) / max(1e-30, float(gd.norm() * r.norm())))
            report[v.key] = dict(rel_l2=e, cos=c)
            if v.key.endswith('weight') and (e > w_rel or c < w_cos):
                bad.append((v.key, report[v.key]))
    return report, bad


def bf16_emulated_step(p0, rnd, real, alpha, cfg, freeze=None, dtype=torch.float64):
    """One `simultaneous` step of the oracle with the bf16 HIP path's storage rounding restated
    (oracle.pgan_oracle.bf16_emulation): the reference the bf16 build is held to tightly."""
    p = {k: v.to(dtype).clone() for k, v in p0.items()}
    with O.bf16_emulation():
        return O.step_simultaneous(p, O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9), None,
                                   {k: v.to(dtype) for k, v in rnd.items()}, real.to(dtype), alpha, cfg, 1e-3, 1e-3,
                                   freeze=freeze)


def bf16_emulation_report(groups, w_rel=0.05, b_rel=0.10):
    """bf16 HIP gradients against the bf16-EMULATING oracle's (same rounding points, masks from the same values; what is
    left is f32 accumulation order and where a fused epilogue rounds a gradient once instead of twice).  Every weight
    gradient within `w_rel` relative L2, every bias gradient within `b_rel` (sums with cancellation).
    groups: [(tag, handle_vars, grads, {name: ref})] -> (report, violations)."""
    report, bad = {}, []
    for tag, hv, grads, refs in groups:
        assert [v.key for v in hv] == list(refs.keys())
        for v, g in zip(hv, grads):
            r = refs[v.key].double()
            gd = g.detach().double().cpu()
            if float(r.norm()) == 0.0:
                assert float(gd.norm()) == 0.0, v.key
                continue
            e = rel_l2(gd, r)
            report[v.key] = round(e, 5)
            if e > (w_rel if v.key.endswith('weight') else b_rel):
                bad.append((v.key, e))
    return report, bad
