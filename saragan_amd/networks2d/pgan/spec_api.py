"""The 2-D networks behind the call signature the step machinery uses (optimization.optimize_step ->
networks.loss.forward_*: generator(z, alpha, phase, base_shape, activation, kernel_spec, filter_spec, param, ...)).
`filter_spec` carries the 2-D tree's legacy triple (variables.legacy_spec); `base_shape` is (C, 1, H0, W0).  Images
cross this interface as D == 1 volumes [N,C,1,H,W]; the gradient penalty reduces over every non-batch axis, as the 2-D
tree's loss does on its 4-D tensors (SURFGAN_2D/networks/loss.py:130-137)."""
from ..ops import as_volume
from . import discriminator as _d
from . import generator as _g


def generator(x, alpha, phase, base_shape, activation, kernel_spec, filter_spec, param=None, size='medium', is_reuse=False,
              conditioning=None):
    if conditioning is not None:
        raise NotImplementedError()
    fs = filter_spec
    img = _g.generator(x, alpha, phase, fs['num_phases'], fs['base_dim'], [base_shape[0], *base_shape[-2:]], activation,
                       param=param, size=fs['size'], is_reuse=is_reuse)
    return as_volume(img)


def discriminator(x, alpha, phase, latent_dim, activation, kernel_spec, filter_spec, param=None, is_reuse=False,
                  conditioning=None):
    if conditioning is not None:
        raise NotImplementedError()
    fs = filter_spec
    return _d.discriminator(x, alpha, phase, fs['num_phases'], fs['base_dim'], latent_dim, activation, param=param,
                            is_reuse=is_reuse, size=fs['size'])


discriminator.sg_gp_full_reduction = True
