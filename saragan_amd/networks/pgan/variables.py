"""Names and shapes of the trainable variables the reference's pgan graph creates at a phase, in creation
order (pgan/generator.py:79-98, pgan/discriminator.py:76-107, ops.py:118,131).  Host
logic only: lets the trainer pre-create variables (flat buffers, checkpoints, the out.txt parameter-count
KAT) without running a kernel."""
from collections import OrderedDict

import numpy as np


def _spec(spec, phase_i, layer_i):
    if phase_i >= len(spec) or layer_i >= len(spec[phase_i]):
        raise ValueError(f'no entry for phase {phase_i}, layer {layer_i} in the kernel/filter spec')
    return spec[phase_i][layer_i]


def pgan_variable_shapes(phase, base_shape, latent_dim, kernel_spec, filter_spec):
    ch = base_shape[0]
    v0 = int(np.prod(base_shape[1:]))
    fs, ks = filter_spec, kernel_spec
    out = OrderedDict()

    def conv(scope, k, cin, cout):
        out[scope + '/weight'] = (*k, cin, cout)
        out[scope + '/bias'] = (cout,)

    def dense(scope, cin, cout):
        out[scope + '/weight'] = (cin, cout)
        out[scope + '/bias'] = (cout,)

    g = 'generator/'
    dense(g + 'generator_in/dense', latent_dim, v0 * _spec(fs, 0, 0))
    conv(g + 'generator_in/conv', _spec(ks, 0, 1), _spec(fs, 0, 0), _spec(fs, 0, 1))
    c_prev = _spec(fs, 0, 1)
    for i in range(2, phase + 1):
        if i == phase:
            conv(g + f'to_rgb_{phase - 1}', (1, 1, 1), c_prev, ch)
        conv(g + f'generator_block_{i}/conv_1', _spec(ks, i - 1, 0), c_prev, _spec(fs, i - 1, 0))
        conv(g + f'generator_block_{i}/conv_2', _spec(ks, i - 1, 1), _spec(fs, i - 1, 0), _spec(fs, i - 1, 1))
        c_prev = _spec(fs, i - 1, 1)
    conv(g + f'to_rgb_{phase}', (1, 1, 1), c_prev, ch)

    d = 'discriminator/'
    conv(d + f'from_rgb_{phase}', (1, 1, 1), ch, _spec(fs, phase - 1, 1))
    c_in = _spec(fs, phase - 1, 1)
    for i in reversed(range(2, phase + 1)):
        conv(d + f'discriminator_block_{i}/conv_1', _spec(ks, i - 1, 1), c_in, _spec(fs, i - 1, 0))
        conv(d + f'discriminator_block_{i}/conv_2', _spec(ks, i - 1, 0), _spec(fs, i - 1, 0), _spec(fs, i - 2, 1))
        c_in = _spec(fs, i - 2, 1)
        if i == phase:
            conv(d + f'from_rgb_{phase - 1}', (1, 1, 1), ch, _spec(fs, phase - 2, 1))
    conv(d + 'discriminator_out', _spec(ks, 0, 1), c_in, _spec(fs, 0, 0))
    dense(d + 'discriminator_out/dense_1', v0 * _spec(fs, 0, 0), latent_dim)
    dense(d + 'discriminator_out/dense_2', latent_dim, 1)
    return out


variable_shapes = pgan_variable_shapes      # every architecture package exposes `variables.variable_shapes`


def preset_specs(size, base_shape, num_phases):
    """kernel_spec / filter_spec equal to the legacy presets: filters from networks/ops.py:201-236, kernel per
    dimension 1 if the extent is < 3 else 3 (networks/ops.py:25-29); reproduces out.txt's parameter counts."""
    from ..ops import k as k_rule, num_filters
    fs, ks = [], []
    for l in range(1, num_phases + 1):
        f = int(num_filters(l, num_phases, base_shape, size=size))
        fs.append([f, f])
        dims = [dd * 2 ** (l - 1) for dd in base_shape[1:]]
        kk = [k_rule(dd) for dd in dims]
        ks.append([kk, kk])
    return ks, fs
