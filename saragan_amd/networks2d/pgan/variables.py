"""Variables of the 2-D pgan at a phase, in creation order (SURFGAN_2D/networks/pgan/generator.py:46-73,
discriminator.py:41-70): conv weights are HWIO [kh,kw,Cin,Cout] (SURFGAN_2D/networks/ops.py:100).  `filter_spec` is
the legacy triple as a dict: {'num_phases', 'base_dim', 'size'} (the 2-D tree has no kernel/filter spec files)."""
from collections import OrderedDict

import numpy as np

from ..ops import k, num_filters


def legacy_spec(num_phases, base_dim, size):
    return dict(num_phases=int(num_phases), base_dim=int(base_dim), size=size)


def variable_shapes(phase, base_shape, latent_dim, kernel_spec, filter_spec):
    """base_shape = (C, 1, H0, W0) (the D == 1 volume form) or (C, H0, W0)."""
    nph, base_dim, size = filter_spec['num_phases'], filter_spec['base_dim'], filter_spec['size']
    ch = base_shape[0]
    h0, w0 = base_shape[-2:]
    out = OrderedDict()

    def conv(scope, kern, cin, cout):
        out[scope + '/weight'] = (*kern, cin, cout)
        out[scope + '/bias'] = (cout,)

    def dense(scope, cin, cout):
        out[scope + '/weight'] = (cin, cout)
        out[scope + '/bias'] = (cout,)

    def kern(level):
        return (k(h0 * 2 ** (level - 1)), k(w0 * 2 ** (level - 1)))

    nf = lambda i: num_filters(i, nph, base_dim, size=size)
    g = 'generator/'
    dense(g + 'generator_in/dense', latent_dim, h0 * w0 * base_dim)
    conv(g + 'generator_in/conv', kern(1), base_dim, base_dim)
    c = base_dim
    for i in range(2, phase + 1):
        if i == phase:
            conv(g + f'to_rgb_{phase - 1}', (1, 1), c, ch)
        conv(g + f'generator_block_{i}/conv_1', kern(i), c, nf(i))
        conv(g + f'generator_block_{i}/conv_2', kern(i), nf(i), nf(i))
        c = nf(i)
    conv(g + f'to_rgb_{phase}', (1, 1), c, ch)
    d = 'discriminator/'
    conv(d + f'from_rgb_{phase}', (1, 1), ch, nf(phase))
    c = nf(phase)
    fout = nf(phase)
    for i in reversed(range(2, phase + 1)):
        conv(d + f'discriminator_block_{i}/conv_1', kern(i), c, nf(i))
        conv(d + f'discriminator_block_{i}/conv_2', kern(i), nf(i), nf(i - 1))
        c = fout = nf(i - 1)
        if i == phase:
            conv(d + f'from_rgb_{phase - 1}', (1, 1), ch, fout)
    conv(d + 'discriminator_out', kern(1), c, fout)
    dense(d + 'discriminator_out/dense_1', h0 * w0 * fout, latent_dim)
    dense(d + 'discriminator_out/dense_2', latent_dim, 1)
    return out
