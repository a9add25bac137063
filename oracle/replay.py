"""CPU replay of the reference's multi-phase training loop (SURFGAN_3D/optuna_objective.py:98-600, normal run) on the
oracle's networks: per-phase variable creation and restore-by-name from the previous phase's checkpoint
(utils.py:75-118), Adam state re-created per phase (:100), alpha schedule (ops.py:4-23, :487-497, :564-570), frozen
previous-phase variables while mixing (quirk Q4, :446-449), EMA after every step (:467) and the end-of-phase overwrite of
the weights by their EMA (quirk Q5, :585-591).  TEST INFRASTRUCTURE ONLY (tests/test_handoff_gpu.py)."""
import torch

from . import pgan_oracle as O


def replay(phases, base_shape, latent, kernel_spec, filter_spec, base_batch_size, mixing_nimg, stabilizing_nimg,
           starting_alpha, lr, ema_beta, loss_fn, gp_weight, noise_stddev, new_variable, batches, randomness,
           dtype=torch.float64):
    """new_variable(name, shape) -> initial value of a variable created at this phase (the product's initialiser order);
    batches(phase, batch_size) -> iterator of real batches; randomness(phase, n, img_shape) -> iterator of dicts with
    z / noise_real / noise_fake / gamma per step.  Returns {phase: weights written to model_{phase}}."""
    ckpt, out = {}, {}
    global_step = 0
    for phase in range(1, phases + 1):
        shapes = O.variable_shapes(phase, base_shape, latent, kernel_spec, filter_spec)
        p = {}
        for name, shp in shapes.items():
            p[name] = ckpt[name].clone() if name in ckpt else new_variable(name, shp).to(dtype)   # restore by NAME
        prev = list(O.variable_shapes(phase - 1, base_shape, latent, kernel_spec, filter_spec).keys()) if phase > 1 else []
        shadow = {k: v.clone() for k, v in p.items()}            # utils.py:106-115: shadows := (restored) weights
        adam_g, adam_d = O.TFAdam(0.0, 0.9), O.TFAdam(0.0, 0.9)  # new graph per phase: fresh Adam slots
        batch_size = max(1, base_batch_size // (2 ** (phase - 1)))
        img = (base_shape[0], *[d * 2 ** (phase - 1) for d in base_shape[1:]])
        cfg = dict(phase=phase, base_shape=base_shape, latent_dim=latent, kernel_spec=kernel_spec, filter_spec=filter_spec,
                   activation='leaky_relu', leakiness=0.2, loss_fn=loss_fn, gp_weight=gp_weight, noise_stddev=noise_stddev)
        alpha = float(starting_alpha) if phase == 1 else 1.0
        mixing = mixing_nimg > 0
        data, rnd = batches(phase, batch_size), randomness(phase, batch_size, img)
        while True:
            real = next(data).to(dtype)
            r = {k: v.to(dtype) for k, v in next(rnd).items()}
            O.step_simultaneous(p, adam_g, adam_d, shadow, r, real, alpha, cfg, lr, lr,
                                freeze=prev if mixing else None, ema_beta=ema_beta)
            global_step += batch_size
            if mixing:
                alpha = O.alpha_update(alpha, mixing_nimg, starting_alpha, batch_size, 1)
            if mixing and global_step >= (phase - 1) * (mixing_nimg + stabilizing_nimg) + mixing_nimg:
                mixing, alpha = False, 0.0
            if global_step >= phase * (mixing_nimg + stabilizing_nimg):
                break
        p = {k: v.clone() for k, v in shadow.items()}            # quirk Q5: weights := EMA, then checkpointed
        ckpt = p
        out[phase] = {k: v.clone() for k, v in p.items()}
    return out
