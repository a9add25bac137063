"""Maps the parameter names of the reference's PyTorch modules (pgan_pytorch/network_dict.py:176-255,
:299-390) onto the TF variable names / layouts of SURFGAN_3D."""
import numpy as np


def oidhw_to_dhwio(w):
    return np.transpose(w, (2, 3, 4, 1, 0))


def discriminator_params(npz, phase):
    """-> {tf_name: ndarray}, and the same mapping for the stored GP gradients."""
    def conv(prefix, tf):
        out[tf + '/weight'] = oidhw_to_dhwio(npz['p:' + prefix + '.weight'])
        out[tf + '/bias'] = npz['p:' + prefix + '.bias']
        if 'gpgrad:' + prefix + '.weight' in npz:
            grads[tf + '/weight'] = oidhw_to_dhwio(npz['gpgrad:' + prefix + '.weight'])
        if 'gpgrad:' + prefix + '.bias' in npz:
            grads[tf + '/bias'] = npz['gpgrad:' + prefix + '.bias']

    def lin(prefix, tf):
        out[tf + '/weight'] = npz['p:' + prefix + '.weight'].T
        out[tf + '/bias'] = npz['p:' + prefix + '.bias']
        if 'gpgrad:' + prefix + '.weight' in npz:
            grads[tf + '/weight'] = npz['gpgrad:' + prefix + '.weight'].T
        if 'gpgrad:' + prefix + '.bias' in npz:
            grads[tf + '/bias'] = npz['gpgrad:' + prefix + '.bias']

    out, grads = {}, {}
    d = 'discriminator/'
    conv('fromrgb_current.fromrgb.0', d + f'from_rgb_{phase}')
    if phase > 1:
        conv('fromrgb_prev.fromrgb.0', d + f'from_rgb_{phase - 1}')
    for i in range(2, phase + 1):
        conv(f'blocks.block_phase_{i}.conv1', d + f'discriminator_block_{i}/conv_1')
        conv(f'blocks.block_phase_{i}.conv2', d + f'discriminator_block_{i}/conv_2')
    conv('discriminator_out.0', d + 'discriminator_out')
    lin('discriminator_out.3', d + 'discriminator_out/dense_1')
    lin('discriminator_out.5', d + 'discriminator_out/dense_2')
    return out, grads


def ref_filter_spec(base_dim, num_phases):
    """pgan_pytorch/network_dict.py:25-28 num_filters -> filter_spec rows [F_l, F_l]."""
    nd = int(np.log2(base_dim / 16))
    f = lambda ph: int(min(base_dim // (2 ** (ph - num_phases + nd)), base_dim))
    return [[f(l), f(l)] for l in range(1, num_phases + 1)]


def generator_p1_params(npz):
    out = {}
    g = 'generator/'
    out[g + 'generator_in/dense/weight'] = npz['p:generator_in.0.weight'].T
    out[g + 'generator_in/dense/bias'] = npz['p:generator_in.0.bias']
    out[g + 'generator_in/conv/weight'] = oidhw_to_dhwio(npz['p:generator_in.3.weight'])
    out[g + 'generator_in/conv/bias'] = npz['p:generator_in.3.bias']
    out[g + 'to_rgb_1/weight'] = oidhw_to_dhwio(npz['p:torgb_current.conv.weight'])
    out[g + 'to_rgb_1/bias'] = npz['p:torgb_current.conv.bias']
    return out


def generator_params(npz, phase):
    """Reference Generator(phase) parameters -> TF names (phase >= 1)."""
    out = generator_p1_params(npz)
    g = 'generator/'
    del out[g + 'to_rgb_1/weight'], out[g + 'to_rgb_1/bias']
    out[g + f'to_rgb_{phase}/weight'] = oidhw_to_dhwio(npz['p:torgb_current.conv.weight'])
    out[g + f'to_rgb_{phase}/bias'] = npz['p:torgb_current.conv.bias']
    if phase > 1:
        out[g + f'to_rgb_{phase - 1}/weight'] = oidhw_to_dhwio(npz['p:torgb_prev.conv.weight'])
        out[g + f'to_rgb_{phase - 1}/bias'] = npz['p:torgb_prev.conv.bias']
    for i in range(2, phase + 1):
        for j in (1, 2):
            out[g + f'generator_block_{i}/conv_{j}/weight'] = oidhw_to_dhwio(npz[f'p:blocks.block_phase_{i}.conv{j}.weight'])
            out[g + f'generator_block_{i}/conv_{j}/bias'] = npz[f'p:blocks.block_phase_{i}.conv{j}.bias']
    return out
