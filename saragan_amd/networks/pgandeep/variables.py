"""Names and shapes of the variables pgandeep creates at a phase, in creation order (pgandeep/generator.py:26-122,
pgandeep/discriminator.py:25-131).  A convolution's input width is whatever reaches it (tf.get_variable sizes the
weight from the tensor, networks/ops.py:148), so the walk below tracks the running channel count."""
from collections import OrderedDict

import numpy as np

from ..specs import filters as F_, kernels as K_


def variable_shapes(phase, base_shape, latent_dim, kernel_spec, filter_spec):
    ch = base_shape[0]
    v0 = int(np.prod(base_shape[1:]))
    fs, ks = filter_spec, kernel_spec
    out = OrderedDict()

    def conv(scope, k, cin, cout):
        out[scope + '/weight'] = (*k, cin, cout)
        out[scope + '/bias'] = (cout,)
        return cout

    def dense(scope, cin, cout):
        out[scope + '/weight'] = (cin, cout)
        out[scope + '/bias'] = (cout,)
        return cout

    if phase > len(ks):
        K_(ks, phase - 1, 0)
    g = 'generator/'
    c = F_(fs, 0, 0)
    dense(g + 'generator_in/dense', latent_dim, v0 * c)
    for j in range(1, len(ks[0])):
        c = conv(g + f'generator_in/conv_{j}', K_(ks, 0, j), c, F_(fs, 0, j))
    for i in range(2, phase + 1):
        if i == phase:
            conv(g + f'to_rgb_{phase - 1}', (1, 1, 1), c, ch)
        for j in range(1, len(ks[i - 1]) + 1):
            c = conv(g + f'generator_block_{i}/conv_{j}', K_(ks, i - 1, j - 1), c, F_(fs, i - 1, j - 1))
    conv(g + f'to_rgb_{phase}', (1, 1, 1), c, ch)

    d = 'discriminator/'
    c = conv(d + f'from_rgb_{phase}', (1, 1, 1), ch, F_(fs, phase - 1, 1))
    for i in reversed(range(2, phase + 1)):
        n = len(ks[i - 1])
        for j in range(1, n + 1):
            nf = F_(fs, i - 2, n - 1) if j == n else F_(fs, i - 1, n - j - 1)
            c = conv(d + f'discriminator_block_{i}/conv_{j}', K_(ks, i - 1, 1), c, nf)
        if i == phase:
            prev = conv(d + f'from_rgb_{phase - 1}', (1, 1, 1), ch, F_(fs, phase - 2, 1))
            if prev != c:
                raise ValueError(f'fade-in needs from_rgb_{phase - 1} ({prev} channels) to match block {i} ({c})')
    n0 = len(ks[0])
    for j in range(1, n0):
        c = conv(d + f'discriminator_out/conv_{j}', K_(ks, 0, n0 - j), c, F_(fs, 0, n0 - j - 1))
    dense(d + 'discriminator_out/dense_1', v0 * c, latent_dim)
    dense(d + 'discriminator_out/dense_2', latent_dim, 1)
    return out
