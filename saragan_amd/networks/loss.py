"""Mirror of SURFGAN_3D/networks/loss.py: forward_generator / forward_discriminator / forward_simultaneous with
the reference's signatures and return tuples.  Random draws go through a RandomSource so that parity tests can
inject z, both noise tensors and gamma (TF and torch RNG streams cannot be matched); by default z/gamma come from
the torch CUDA generator and the instance noise from the library's Philox kernel (sg_add_noise)."""
import os

import torch
import torch.nn.functional as TF

from .. import functional as F
from ..varstore import compute_dtype, current_store


class RandomSource:
    """Default randomness: device-side draws.  Counters make every call use fresh Philox offsets."""

    def __init__(self, seed=0, device='cuda'):
        self.seed = int(seed)
        self.calls = 0
        self.gen = torch.Generator(device=device).manual_seed(self.seed)

    def latent(self, n, latent_dim, device):
        return torch.randn(n, latent_dim, device=device, generator=self.gen)

    def gamma(self, n, device):
        return torch.rand(n, 1, 1, 1, 1, device=device, generator=self.gen)

    def add_noise(self, x, stddev, tag):
        self.calls += 1
        return F.add_noise(x, stddev, self.seed, offset=self.calls << 40)


_DEVICE_RANDOM = RandomSource      # (the class itself: tests substitute `RandomSource` with host-drawn variants)


def graph_safe(src):
    """Whether a captured step (loss.StaticRandom) can stand in for `src`: the device-side source with none of its draws
    overridden."""
    return (type(src) is _DEVICE_RANDOM or (isinstance(src, _DEVICE_RANDOM) and all(
        getattr(type(src), m) is getattr(_DEVICE_RANDOM, m) for m in ('latent', 'gamma', 'add_noise')))) and \
        hasattr(src, 'calls') and hasattr(src, 'gen')


class InjectedRandom(RandomSource):
    """Replays stored tensors: keys z, noise_real, noise_fake, gamma (oracle.pgan_oracle.draw_randomness)."""

    def __init__(self, tensors):
        self.t = tensors

    def latent(self, n, latent_dim, device):
        return self.t['z'].to(device)

    def gamma(self, n, device):
        return self.t['gamma'].to(device)

    def add_noise(self, x, stddev, tag):
        noise = self.t[tag].to(x.device, x.dtype)
        return F.lerp(x, noise.contiguous(memory_format=torch.channels_last_3d), 1.0, float(stddev))


class StaticRandom:
    """The randomness of a CAPTURED training step (optimization.StepGraph, SARAGAN_HIPGRAPH=1): latents and the gradient
    penalty's mixing weights are drawn from `base`'s generator OUTSIDE the captured region into fixed buffers (`draw()`, same
    order and shapes as the eager step: bit-identical values), the instance noise reads its Philox offset from a device
    counter that the captured launches advance themselves.  `after_replay()` keeps `base.calls` in step, so that an eager
    step that follows continues the same sequence."""

    def __init__(self, base, n, latent_dim, device, pattern=('z', 'g')):
        # pattern: the step's draws in the order the eager step makes them ('z': latents, 'g': mixing weights) -- ('z', 'g') for a
        # `simultaneous` step, ('z', 'g', 'z') for `alternate` (forward_discriminator, then forward_generator draws its own latents)
        self.base, self.n, self.latent_dim = base, int(n), int(latent_dim)
        self.pattern = tuple(pattern)
        self.zs = [torch.empty(self.n, self.latent_dim, device=device) for k in self.pattern if k == 'z']
        self.gs = [torch.empty(self.n, 1, 1, 1, 1, device=device) for k in self.pattern if k == 'g']
        self.z, self.g = self.zs[0], self.gs[0]
        self._iz = self._ig = 0
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)
        self.noise_calls = 0         # add_noise calls of one step (counted while capturing)
        self.counting = False

    def draw(self):
        # (straight into the fixed buffers: the same draws from the same generator as RandomSource.latent / .gamma)
        iz = ig = 0
        for k in self.pattern:
            if k == 'z':
                torch.randn(self.n, self.latent_dim, generator=self.base.gen, out=self.zs[iz])
                iz += 1
            else:
                torch.rand(self.n, 1, 1, 1, 1, generator=self.base.gen, out=self.gs[ig])
                ig += 1
        self._iz = self._ig = 0

    def sync_counter(self):
        if getattr(self, '_synced_calls', None) != self.base.calls:      # (only after eager steps moved the host count on)
            self.counter.fill_((self.base.calls + 1) << 40)

    def latent(self, n, latent_dim, device):
        assert (int(n), int(latent_dim)) == (self.n, self.latent_dim)
        z = self.zs[min(self._iz, len(self.zs) - 1)]
        self._iz += 1
        return z

    def gamma(self, n, device):
        assert int(n) == self.n
        g = self.gs[min(self._ig, len(self.gs) - 1)]
        self._ig += 1
        return g

    def add_noise(self, x, stddev, tag):
        if self.counting:
            self.noise_calls += 1
        return F.add_noise(x, stddev, self.base.seed, offset=self.counter)

    def after_replay(self):
        self.base.calls += self.noise_calls
        self._synced_calls = self.base.calls      # the device counter moved on by the same amount


_RANDOM = {'src': None}
_LINK = {'on': False}
_NO_BATCHED_D = bool(int(os.environ.get('SARAGAN_NO_BATCHED_D', '0')))   # diagnostic: separate D(real) / D(fake) passes


class linear_generator_link:
    """Context for callers that run BOTH backward passes of forward_simultaneous themselves (optimization.StepGraph).
    With the wgan loss d gen_loss / d D(fake) = -1/N = -(d disc_loss / d D(fake)) element for element, and the
    discriminator's data-gradient chain is linear in its upstream gradient, so the gradient of gen_loss at the
    discriminator's INPUT is exactly minus what the disc_loss backward delivers there.  Inside this context the fake
    batch enters D through a detached leaf and gen_loss carries `.sg_link = (gen_sample_noisy, leaf, -1.0)`: the
    caller back-propagates disc_loss to `leaf` along with D's variables and then starts the generator's backward at
    gen_sample_noisy with factor * leaf.grad — one whole data-gradient pass through D less per step
    (optimization.py:128-163 computes the two tf.gradients separately).  Only exact for a uniform factor (wgan);
    the logistic loss keeps the general path."""

    def __enter__(self):
        self.prev = _LINK['on']
        _LINK['on'] = True

    def __exit__(self, *a):
        _LINK['on'] = self.prev


def set_random_source(src):
    _RANDOM['src'] = src


def _rng(device):
    if _RANDOM['src'] is None:
        _RANDOM['src'] = RandomSource(0, device)
    return _RANDOM['src']


def _img(x):
    """real images -> compute dtype, NDHWC."""
    return x.to(compute_dtype()).contiguous(memory_format=torch.channels_last_3d)


def forward_generator(generator, discriminator, real_image_input, latent_dim, alpha, phase, base_shape,
                      kernel_spec, filter_spec, activation, leakiness, loss_fn, noise_stddev, is_reuse=False):
    """networks/loss.py:4-39."""
    rng = _rng(real_image_input.device)
    z = rng.latent(real_image_input.shape[0], latent_dim, real_image_input.device)
    gen_sample = generator(z, alpha, phase, base_shape, activation=activation, kernel_spec=kernel_spec,
                           filter_spec=filter_spec, param=leakiness, is_reuse=is_reuse)
    real_image_input = rng.add_noise(_img(real_image_input), noise_stddev, 'noise_real')  # drawn as in the reference
    gen_sample_noisy = rng.add_noise(gen_sample, noise_stddev, 'noise_fake')
    disc_fake_g = discriminator(gen_sample_noisy, alpha, phase, latent_dim=latent_dim, activation=activation,
                                kernel_spec=kernel_spec, filter_spec=filter_spec, param=leakiness,
                                is_reuse=is_reuse).float()
    if loss_fn == 'wgan':
        gen_loss = -torch.mean(disc_fake_g)
    elif loss_fn == 'logistic':
        gen_loss = torch.mean(TF.softplus(-disc_fake_g))
    else:
        raise ValueError(f"Unknown loss function: {loss_fn}")
    return gen_sample, gen_loss


def _gradient_slopes_sq(discriminator, interpolates, alpha, phase, latent_dim, activation, kernel_spec, filter_spec,
                        leakiness, keep_w):
    """d D(x)/dx at the interpolates with a differentiable graph (tf.gradients at loss.py:74-77 / :136-139),
    then sum of squares over (c,d,h) [keep_w, quirk Q1] or over (c,d,h,w)."""
    interpolates = interpolates.detach().requires_grad_(True)
    d_int = discriminator(interpolates, alpha, phase, latent_dim=latent_dim, is_reuse=True, activation=activation,
                          kernel_spec=kernel_spec, filter_spec=filter_spec, param=leakiness)
    with F.skip_param_grads(p for _, p in current_store().trainable('discriminator/')):   # only d/dx is asked for
        (gradients,) = torch.autograd.grad(d_int, interpolates, grad_outputs=torch.ones_like(d_int),
                                           create_graph=True)
    ss = F.sumsq_keep_w(gradients)          # [N, W] f32
    return ss if keep_w else ss.sum(dim=1)


def forward_discriminator(generator, discriminator, real_image_input, latent_dim, alpha, phase, base_shape,
                          kernel_spec, filter_spec, activation, leakiness, loss_fn, gp_weight, noise_stddev,
                          is_reuse=False):
    """networks/loss.py:42-98 (gradient penalty over axes (1,2,3,4), loss.py:79)."""
    rng = _rng(real_image_input.device)
    z = rng.latent(real_image_input.shape[0], latent_dim, real_image_input.device)
    with torch.no_grad():   # every use of gen_sample in this function sits behind tf.stop_gradient
        gen_sample = generator(z, alpha, phase, base_shape, activation=activation, kernel_spec=kernel_spec,
                               filter_spec=filter_spec, param=leakiness, is_reuse=is_reuse)
    real_image_input = rng.add_noise(_img(real_image_input), noise_stddev, 'noise_real')
    gen_sample_noisy = rng.add_noise(gen_sample, noise_stddev, 'noise_fake')
    net = dict(latent_dim=latent_dim, activation=activation, kernel_spec=kernel_spec, filter_spec=filter_spec,
               param=leakiness)
    disc_fake_d = discriminator(gen_sample_noisy.detach(), alpha, phase, **net).float()
    disc_real = discriminator(real_image_input, alpha, phase, is_reuse=True, **net).float()
    gamma = rng.gamma(real_image_input.shape[0], real_image_input.device)
    interpolates = F.interpolate_rows(gamma, real_image_input, gen_sample_noisy)      # f32 weights, one rounding
    slopes = torch.sqrt(_gradient_slopes_sq(discriminator, interpolates, alpha, phase, latent_dim, activation,
                                            kernel_spec, filter_spec, leakiness, keep_w=False))
    if loss_fn == 'wgan':
        gradient_penalty = (slopes - 1) ** 2
        gp_loss = gp_weight * gradient_penalty
        disc_loss = disc_fake_d - disc_real
        drift_loss = 1e-3 * disc_real ** 2
        disc_loss = torch.mean(disc_loss + gp_loss + drift_loss)   # [N,1] + [N] broadcasts as in TF
    elif loss_fn == 'logistic':
        gradient_penalty = torch.mean(slopes ** 2)
        gp_loss = gp_weight * gradient_penalty
        disc_loss = torch.mean(TF.softplus(disc_fake_d)) + torch.mean(TF.softplus(-disc_real))
        disc_loss = disc_loss + gp_loss
    else:
        raise ValueError(f"Unknown loss function: {loss_fn}")
    return disc_loss, gp_loss


def forward_simultaneous(generator, discriminator, real_image_input, latent_dim, alpha, phase, base_shape,
                         kernel_spec, filter_spec, activation, leakiness, loss_fn, gp_weight, noise_stddev,
                         conditioning=None):
    """networks/loss.py:101-165, including quirk Q1 (slopes keeps the W axis: loss.py:140)."""
    rng = _rng(real_image_input.device)
    z = rng.latent(real_image_input.shape[0], latent_dim, real_image_input.device)
    gen_sample = generator(z, alpha, phase, base_shape, activation=activation, kernel_spec=kernel_spec,
                           filter_spec=filter_spec, param=leakiness, conditioning=conditioning)
    real_image_input = rng.add_noise(_img(real_image_input), noise_stddev, 'noise_real')
    gen_sample_noisy = rng.add_noise(gen_sample, noise_stddev, 'noise_fake')
    net = dict(latent_dim=latent_dim, activation=activation, kernel_spec=kernel_spec, filter_spec=filter_spec,
               param=leakiness, conditioning=conditioning)
    # The reference evaluates D(stop_gradient(fake)) for the D loss and D(fake) for the G loss (loss.py:126-128,
    # :143-144): the same forward values.  ONE pass serves both here: the D-loss gradient is only taken w.r.t. D's
    # variables (the edge into G is never followed), the G-loss gradient only w.r.t. G's (no D weight gradients).
    link = _LINK['on'] and loss_fn == 'wgan' and gen_sample_noisy.requires_grad
    if link and not _NO_BATCHED_D:
        # The pgan discriminator has no op that couples batch samples (its minibatch-stddev layer is disabled,
        # pgan/discriminator.py:50), so D(real) and D(fake) are ONE pass over the concatenated batch: a third fewer
        # launches in D's forward and backward, twice the tiles for the low-resolution layers.
        fake_in = gen_sample_noisy.detach().requires_grad_(True)
        n_real = real_image_input.shape[0]
        both = discriminator(torch.cat([real_image_input, fake_in], dim=0), alpha, phase, **net).float()
        disc_real, disc_fake_g = both[:n_real], both[n_real:]
        disc_fake_d = disc_fake_g
    else:
        fake_in = gen_sample_noisy.detach().requires_grad_(True) if link else gen_sample_noisy
        disc_fake_g = discriminator(fake_in, alpha, phase, **net).float()
        disc_fake_d = disc_fake_g
        disc_real = discriminator(real_image_input, alpha, phase, is_reuse=True, **net).float()
    gamma = rng.gamma(real_image_input.shape[0], real_image_input.device)
    interpolates = F.interpolate_rows(gamma, real_image_input, gen_sample_noisy)      # f32 weights, one rounding
    # quirk Q1 belongs to the 3-D tree's 5-D tensors; the 2-D tree reduces its 4-D gradient over every non-batch axis
    keep_w = not getattr(discriminator, 'sg_gp_full_reduction', False)
    slopes = torch.sqrt(_gradient_slopes_sq(discriminator, interpolates, alpha, phase, latent_dim, activation,
                                            kernel_spec, filter_spec, leakiness, keep_w=keep_w))
    if loss_fn == 'wgan':
        gradient_penalty = (slopes - 1) ** 2
        if not keep_w:
            gradient_penalty = gradient_penalty.reshape(-1, 1)     # per sample, next to D's [N,1] logits
        gp_loss = gp_weight * gradient_penalty
        disc_loss = disc_fake_d - disc_real
        drift_loss = 1e-3 * disc_real ** 2
        disc_loss = torch.mean(disc_loss + gp_loss + drift_loss)
        gen_loss = -torch.mean(disc_fake_g)
        if link:
            gen_loss.sg_link = (gen_sample_noisy, fake_in, -1.0)
    elif loss_fn == 'logistic':
        gradient_penalty = torch.mean(slopes ** 2)
        gp_loss = gp_weight * gradient_penalty
        disc_loss = torch.mean(TF.softplus(disc_fake_d)) + torch.mean(TF.softplus(-disc_real))
        disc_loss = disc_loss + gp_loss
        gen_loss = torch.mean(TF.softplus(-disc_fake_g))
    else:
        raise ValueError(f"Unknown loss function: {loss_fn}")
    return gen_loss, disc_loss, gp_loss, gen_sample
