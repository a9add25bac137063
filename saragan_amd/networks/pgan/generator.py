"""Mirror of SURFGAN_3D/networks/pgan/generator.py (same callables, argument order, variable scopes and
ValueError / NotImplementedError behaviour); the ops are saragan_amd.networks.ops (gfx950 kernels)."""
import numpy as np

from ..ops import *  # noqa: F401,F403  (the reference does `from networks.ops import *`)
from ..ops import (act, apply_bias, conv3d, dense, lerp, materialize, pixel_norm, to_rgb, upscale3d,
                   variable_scope)


def get_filters_generator(filter_spec, phase_i, layer_i):
    """pgan/generator.py:4-13."""
    if phase_i >= len(filter_spec):
        print(f"Error: no filter count specified for phase {phase_i}. Please check the file passed to --filter_spec.")
        raise ValueError
    if layer_i >= len(filter_spec[phase_i]):
        print(f"Error: no filter count specified for layer {layer_i} in phase {phase_i}. Please check the file passed to --filter_spec.")
        raise ValueError
    return filter_spec[phase_i][layer_i]


def get_kernels_generator(kernel_spec, phase_i, layer_i):
    """pgan/generator.py:15-24."""
    if phase_i >= len(kernel_spec):
        print(f"Error: no kernel shape specified for phase {phase_i}. Please check the file passed to --kernel_spec.")
        raise ValueError
    if layer_i >= len(kernel_spec[phase_i]):
        print(f"Error: no kernel shape specified for layer {layer_i} in phase {phase_i}. Please check the file passed to --kernel_spec.")
        raise ValueError
    return kernel_spec[phase_i][layer_i]


def generator_in(x, shape, activation, kernel_spec, filter_spec, param=None):
    """pgan/generator.py:26-45."""
    with variable_scope('dense'):
        x = dense(x, int(np.prod(shape)) * get_filters_generator(filter_spec, 0, 0), activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
    x = materialize(x).reshape([-1, get_filters_generator(filter_spec, 0, 0)] + list(shape))
    with variable_scope('conv'):
        x = conv3d(x, get_filters_generator(filter_spec, 0, 1), get_kernels_generator(kernel_spec, 0, 1), activation,
                   param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
        x = pixel_norm(x)
    return x


def generator_block(x, activation, kernel_spec, filter_spec, i, param=None):
    """pgan/generator.py:48-71."""
    with variable_scope('upsample'):
        x = upscale3d(x)
    with variable_scope('conv_1'):
        kernel = get_kernels_generator(kernel_spec, i - 1, 0)
        x = conv3d(x, get_filters_generator(filter_spec, i - 1, 0), kernel, activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
        x = pixel_norm(x)
    with variable_scope('conv_2'):
        kernel = get_kernels_generator(kernel_spec, i - 1, 1)
        x = conv3d(x, get_filters_generator(filter_spec, i - 1, 1), kernel, activation, param=param)
        x = apply_bias(x)
        x = act(x, activation, param=param)
        x = pixel_norm(x)
    return x


def generator(x, alpha, phase, base_shape, activation, kernel_spec, filter_spec, param=None, size='medium',
              is_reuse=False, conditioning=None):
    """pgan/generator.py:74-103."""
    if conditioning is not None:
        raise NotImplementedError()
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
