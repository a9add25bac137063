"""Mirror of SURFGAN_3D/networks/pgandeep/generator.py: pgan's generator with N = len(kernel_spec[phase])
convolutions per block (pgandeep/generator.py:26-122).  Same signature as pgan's, so optimize_step takes either
(SURVEY.md section 2a #14); variable scopes follow the reference: `generator_in/conv_{j}` (j = 1..N0-1),
`generator_block_{i}/conv_{j}` (j = 1..Ni)."""
import numpy as np

from ..ops import *  # noqa: F401,F403  (the reference does `from networks.ops import *`)
from ..ops import act, apply_bias, conv3d, dense, lerp, materialize, pixel_norm, to_rgb, upscale3d, variable_scope
from ..specs import filters, kernels


def _conv_stage(x, nfilters, kernel, activation, param):
    x = conv3d(x, nfilters, kernel, activation, param=param)
    x = apply_bias(x)
    x = act(x, activation, param=param)
    return pixel_norm(x)


def generator_in(x, shape, activation, kernel_spec, filter_spec, param=None):
    """pgandeep/generator.py:26-56: the dense layer stands for layer 0 of phase 0, so N0 - 1 convolutions follow."""
    with variable_scope('dense'):
        x = dense(x, int(np.prod(shape)) * filters(filter_spec, 0, 0), activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
    x = materialize(x).reshape([-1, filters(filter_spec, 0, 0)] + list(shape))
    for layer_i in range(1, len(kernel_spec[0])):
        with variable_scope(f'conv_{layer_i}'):
            x = _conv_stage(x, filters(filter_spec, 0, layer_i), kernels(kernel_spec, 0, layer_i), activation, param)
    return x


def generator_block(x, activation, kernel_spec, filter_spec, i, param=None):
    """pgandeep/generator.py:59-94."""
    with variable_scope('upsample'):
        x = upscale3d(x)
    for layer_i in range(1, len(kernel_spec[i - 1]) + 1):
        with variable_scope(f'conv_{layer_i}'):
            x = _conv_stage(x, filters(filter_spec, i - 1, layer_i - 1), kernels(kernel_spec, i - 1, layer_i - 1),
                            activation, param)
    return x


def generator(x, alpha, phase, base_shape, activation, kernel_spec, filter_spec, param=None, size='medium',
              is_reuse=False, conditioning=None):
    """pgandeep/generator.py:97-122."""
    if conditioning is not None:
        raise NotImplementedError()
    if phase > len(kernel_spec):
        kernels(kernel_spec, phase - 1, 0)      # raises the reference's ValueError for a missing phase
    with variable_scope('generator', reuse=is_reuse):
        with variable_scope('generator_in'):
            x = generator_in(x, shape=base_shape[1:], activation=activation, kernel_spec=kernel_spec,
                             filter_spec=filter_spec, param=param)
        x_upsample = None
        for i in range(2, phase + 1):
            if i == phase:
                with variable_scope(f'to_rgb_{phase - 1}'):
                    x_upsample = upscale3d(to_rgb(x, channels=base_shape[0]))
            with variable_scope(f'generator_block_{i}'):
                x = generator_block(x, activation=activation, kernel_spec=kernel_spec, filter_spec=filter_spec, i=i,
                                    param=param)
        with variable_scope(f'to_rgb_{phase}'):
            x_out = to_rgb(x, channels=base_shape[0])
        if x_upsample is not None:
            x_out = lerp(x_upsample, x_out, alpha)     # alpha * x_upsample + (1 - alpha) * x_out
        return materialize(x_out)
