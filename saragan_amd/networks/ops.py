"""Mirror of SURFGAN_3D/networks/ops.py: same function names, argument order and error behaviour, with the
TF1 ops replaced by the gfx950 kernels behind include/saragan_hip.h (via saragan_amd.functional).

Tensors are NCDHW-shaped torch tensors on the GPU (stored channels-last).  conv3d / upscale3d return lazy
handles so that the reference's sequence `conv3d -> apply_bias -> act -> pixel_norm` (pgan/generator.py:40-44)
and `upscale3d -> conv3d` (pgan/generator.py:49-57) collapse into ONE kernel launch while the network code
keeps the reference's op-by-op structure.  Any other consumer materialises the handle.
"""
import os
import numpy as np
import torch

from .. import functional as F
from ..varstore import compute_dtype, get_variable, variable_scope  # noqa: F401  (re-exported)


# ---- lazy handles ----------------------------------------------------------------------------------
class _LazyUp:
    """upscale3d(x) not yet materialised: a following conv3d reads x through a nearest-x2 gather."""

    def __init__(self, x):
        self._src = x          # a tensor or a lazy handle: evaluated only when somebody reads the up-scaled tensor, so a
        self._v = None         # branch that lerp() prunes (alpha = 0) never runs the to_rgb / from_rgb it hangs on

    @property
    def x(self):
        if not isinstance(self._src, torch.Tensor):
            self._src = _val(self._src)
        return self._src

    @property
    def shape(self):
        n, c, d, h, w = self._src.shape
        return torch.Size((n, c, 2 * d, 2 * h, 2 * w))

    def value(self):
        if self._v is None:
            self._v = F.upscale2x(self.x, 1.0)
        return self._v


class _LazyConv:
    """conv3d(x, w) with a growing epilogue (bias -> act -> pixel_norm, in that order only)."""

    def __init__(self, x, w, coef, ups):
        self.x, self.w, self.coef, self.ups = x, w, coef, ups
        self.bias = None
        self.act = False
        self.slope = 0.2
        self.pn = False
        self.eps = 1e-8
        self.stage = 0   # 0 conv, 1 +bias, 2 +act, 3 +pixel_norm
        self.in_info = None    # ActInfo of the LeakyReLU output this conv consumes (fused mask in my data gradient)
        self.act_info = None   # ActInfo of MY output when it ends in a LeakyReLU (and no pixel-norm)
        self._v = None

    @property
    def shape(self):
        n = self.x.shape[0]
        sp = [2 * s for s in self.x.shape[2:]] if self.ups else list(self.x.shape[2:])
        return torch.Size((n, self.w.shape[-1], *sp))

    def value(self):
        if self._v is None:
            out_info = self.act_info if self.act else None
            self._v = F.conv3d(self.x, self.w, self.coef, bias=self.bias, act=self.act, slope=self.slope,
                               pixel_norm=self.pn, eps=self.eps, upsample_in=self.ups, out_info=out_info,
                               in_info=self.in_info)
        return self._v


class _LazyRgbTail(_LazyConv):
    """to_rgb applied to a generator stage that has not been materialised yet (conv -> bias -> LeakyReLU -> pixel_norm):
    evaluated together with it as one autograd node (functional._ConvPnActToRgb), whose backward never writes to_rgb's
    full-resolution data gradient.  The stage's own handle receives its value too, so later consumers reuse it."""

    def __init__(self, prod, w, coef):
        super().__init__(None, w, coef, False)
        self.prod = prod

    @property
    def shape(self):
        return torch.Size((self.prod.shape[0], self.w.shape[-1], *self.prod.shape[2:]))

    def value(self):
        if self._v is None:
            p = self.prod
            if p._v is None and not self.act and not self.pn:
                p.act_info = None      # this node applies the stage's backward itself: no consumer may register for it
                self._v, p._v = F.conv3d_pn_to_rgb(p.x, p.w, p.coef, p.bias, p.ups, p.slope, p.eps, p.in_info,
                                                   self.w, self.coef, self.bias)
            else:      # the stage was materialised by another consumer first, or the tail grew an epilogue of its own
                # (registered as an ordinary consumer: an earlier consumer may have signed up for the stage's backward)
                self._v = F.conv3d(_consume(p, False)[0], self.w, self.coef, bias=self.bias, act=self.act, slope=self.slope,
                                   pixel_norm=self.pn, eps=self.eps)
        return self._v


def _consume(x, premask=False, pn_ok=False):
    """Materialises a handle for one consumer and returns (tensor, ActInfo-or-None).  `premask` says that this
    consumer applies the producer's LeakyReLU-backward mask inside its own backward kernel (F.ActInfo); for a stage that
    goes on through pixel_norm it must also be able to apply that backward (`pn_ok`, F.pn_bwd_epilogue_available)."""
    if isinstance(x, _LazyConv):
        info = x.act_info if x.act else None
        if x.pn:
            premask = premask and pn_ok
        if info is not None:
            info.consume(premask)
        return x.value(), (info if premask else None)
    if isinstance(x, _LazyUp):
        return x.value(), None
    return x, None


def _val(x):
    return _consume(x, False)[0]


# ---- reference functions -----------------------------------------------------------------------------
class ScalarVariable:
    """Host-side stand-in for a non-trainable scalar tf.Variable (alpha, learning rates, step counters)."""

    def __init__(self, value, name='', dtype=np.float32):
        self.name = name
        self.dtype = dtype
        self.value = dtype(value)

    def assign(self, v):
        self.value = self.dtype(v)
        return self.value

    def eval(self):
        return self.value

    def __float__(self):
        return float(self.value)


class Op:
    """A deferred host action (the eager counterpart of a tf.Operation); run with .run() or Session.run."""

    def __init__(self, fn, name=''):
        self.fn, self.name = fn, name

    def run(self):
        return self.fn()


def alpha_update(alpha, mixing_nimg, starting_alpha, batch_size, global_size):
    """networks/ops.py:4-23: returns the op that lowers alpha by starting_alpha/num_steps (floored at 0)."""
    if mixing_nimg == 0:
        return Op(lambda: alpha.assign(0), 'alpha_update')
    num_steps = mixing_nimg // (batch_size * global_size)
    upd = np.float32(starting_alpha / num_steps)
    return Op(lambda: alpha.assign(max(np.float32(alpha.value) - upd, np.float32(0))), 'alpha_update')


def k(x):
    """networks/ops.py:25-29."""
    return 1 if x < 3 else 3


def get_kernel(x_shape, desired_k_shape):
    """networks/ops.py:31-58."""
    if len(x_shape) != len(desired_k_shape):
        print(f"x_shape: {x_shape}. desired_k_shape: {desired_k_shape}")
    assert len(x_shape) == len(desired_k_shape)
    kernel = []
    for x_i, k_i in zip(x_shape, desired_k_shape):
        if x_i < k_i:
            kernel.append(x_i - 1 if (x_i % 2) == 0 else x_i)
        else:
            kernel.append(k_i)
    return kernel


def calculate_gain(activation, param=None):
    """networks/ops.py:60-77."""
    linear_fns = ['linear', 'conv1d', 'conv2d', 'conv3d', 'conv_transpose1d', 'conv_transpose2d', 'conv_transpose3d']
    if activation in linear_fns or activation == 'sigmoid':
        return 1
    elif activation == 'tanh':
        return 5.0 / 3
    elif activation == 'relu':
        return np.sqrt(2.0)
    elif activation == 'leaky_relu':
        assert param is not None
        if not isinstance(param, bool) and isinstance(param, int) or isinstance(param, float):
            negative_slope = param
        else:
            raise ValueError("negative_slope {} not a valid number".format(param))
        return np.sqrt(2.0 / (1 + negative_slope ** 2))
    else:
        raise ValueError("Unsupported nonlinearity {}".format(activation))


class ScaledWeight:
    """`w * runtime_coef` of networks/ops.py:121-122 kept factored: the raw f32 variable (DHWIO / [in,out])
    plus the scalar the weight-pack kernel applies."""

    def __init__(self, var, coef):
        self.var, self.coef = var, float(coef)

    @property
    def shape(self):
        return self.var.shape


def get_weight(shape, activation, lrmul=1, use_eq_lr=True, use_spectral_norm=False, param=None):
    """networks/ops.py:111-127."""
    fan_in = np.prod(shape[:-1])
    gain = calculate_gain(activation, param)
    he_std = gain / np.sqrt(fan_in)
    runtime_coef = he_std * lrmul
    if use_spectral_norm:
        raise NotImplementedError('spectral_norm is not used by the pgan path')
    from ..varstore import current_store, scope_name
    full = (scope_name() + '/weight') if scope_name() else 'weight'
    fresh = full not in current_store().vars
    w = get_variable('weight', shape, 'normal')
    if fresh and lrmul != 1:      # init_std = 1 / lrmul (ops.py:115,118-119); the coefficient below carries lrmul back in
        with torch.no_grad():
            w.mul_(1.0 / lrmul)
    return ScaledWeight(w, runtime_coef if use_eq_lr else 1.0)


def apply_bias(x, lrmul=1):
    """networks/ops.py:130-136."""
    b = get_variable('bias', [x.shape[1]], 'zeros')
    if lrmul != 1:                # ops.py:131: the variable times lrmul (a [C]-sized torch op; pgan never passes it)
        b = b * float(lrmul)
    if isinstance(x, _LazyConv) and x.stage == 0 and x._v is None:
        x.bias, x.stage = b, 1
        return x
    return F.bias_act(_val(x), b, False, 0.0)


def dense(x, fmaps, activation, lrmul=1, param=None):
    """networks/ops.py:139-144 (flatten is C-major of NCDHW, as tf.reshape on the reference's tensors)."""
    x = _val(x)
    if len(x.shape) > 2:
        x = x.reshape(x.shape[0], int(np.prod(x.shape[1:])))
    w = get_weight([x.shape[1], fmaps], activation, lrmul=lrmul, param=param)
    # the discriminator's logit layer (one output unit) runs in f32 whatever the compute dtype: the losses add terms
    # 1000x apart (wgan drift 1e-3 * D(real)^2 next to D(fake) - D(real), loss.py:146-151), below bf16 resolution
    dt = torch.float32 if fmaps == 1 else compute_dtype()
    return _LazyConv(x.to(dt), w.var, w.coef, False)


def conv3d(x, fmaps, kernel, activation, param=None, lrmul=1):
    """networks/ops.py:147-150."""
    ups = isinstance(x, _LazyUp) and x._v is None
    cin = x.shape[1]
    if (tuple(kernel) == (1, 1, 1) and fmaps <= 4 and type(x) is _LazyConv and x._v is None and x.stage == 3 and x.act and
            x.pn and x.x.dtype == compute_dtype() and not F._NO_RGB_FUSION):
        w = get_weight([*kernel, cin, fmaps], activation, param=param, lrmul=lrmul)
        return _LazyRgbTail(x, w.var, w.coef)
    if ups:
        xin, in_info = x.x, None
    else:
        pn_ok = isinstance(x, _LazyConv) and x.pn and F.pn_bwd_epilogue_available(tuple(x.shape), kernel, fmaps, compute_dtype())
        xin, in_info = _consume(x, premask=True, pn_ok=pn_ok)
    w = get_weight([*kernel, cin, fmaps], activation, param=param, lrmul=lrmul)
    if xin.dtype != compute_dtype():
        xin, in_info = xin.to(compute_dtype()), None
    lz = _LazyConv(xin, w.var, w.coef, ups)
    lz.in_info = in_info
    return lz


def leaky_relu(x, alpha_lr=0.2):
    """networks/ops.py:167-182 (mask taken from the output, subgradient 1 at 0)."""
    if isinstance(x, _LazyConv) and x.stage <= 1 and x._v is None:
        x.act, x.slope, x.stage = True, float(alpha_lr), 2
        x.act_info = F.ActInfo(alpha_lr)
        return x
    return F.bias_act(_val(x), None, True, float(alpha_lr))


def act(x, activation, param=None):
    """networks/ops.py:185-192."""
    if activation == 'leaky_relu':
        assert param is not None
        return leaky_relu(x, alpha_lr=param)
    elif activation == 'linear':
        return x
    else:
        raise ValueError(f"Unknown activation {activation}")


def num_filters(phase, num_phases, base_shape, base_dim=None, size=None):
    """networks/ops.py:201-236."""
    lists = {
        'xxs': [256, 256, 64, 32, 16, 8, 4, 2], 'xs': [256, 256, 64, 64, 32, 16, 8, 4],
        's': [512, 512, 128, 128, 64, 32, 16, 8], 'm': [1024, 1024, 256, 256, 128, 64, 32, 16],
        'l': [2048, 2048, 512, 512, 256, 128, 64, 32], 'xl': [4096, 4096, 1024, 1024, 512, 256, 128, 64],
        'xxl': [8192, 8192, 2048, 1024, 1024, 512, 256, 128]}
    if size not in lists:
        raise ValueError(f"Unknown size: {size}")
    filter_list = lists[size]
    assert len(filter_list) == 8, "Filter lists are built for LIDC-IDRI dataset."
    current_dim = [2 ** (phase - 1) * dim for dim in base_shape[1:]]
    log_product = np.log2(np.prod(current_dim))
    reference_log = [4 + n * 3 for n in range(0, 7)]
    index = np.argmin(np.abs(np.array(reference_log) - log_product))
    return filter_list[index]


def to_rgb(x, channels=1):
    """networks/ops.py:239-240."""
    return apply_bias(conv3d(x, channels, (1, 1, 1), activation='linear'))


def from_rgb(x, filters_out, activation, param=None):
    """networks/ops.py:243-247."""
    x = conv3d(x, filters_out, (1, 1, 1), activation, param)
    x = apply_bias(x)
    x = act(x, activation, param=param)
    return x


def avg_unpool3d(x, factor=2, gain=1):
    """networks/ops.py:250-262."""
    x = _val(x)
    if factor == 1:
        return x if gain == 1 else F.lerp(x, None, float(gain), 0.0)
    if factor != 2:
        raise NotImplementedError('only factor 2 is used by the pgan path')
    return F.upscale2x(x, float(gain))


def avg_pool3d(x, factor=2, gain=1):
    """networks/ops.py:265-273."""
    if factor == 1:
        x = _val(x)
        return x if gain == 1 else F.lerp(x, None, float(gain), 0.0)
    if factor != 2:
        raise NotImplementedError('only factor 2 is used by the pgan path')
    if (gain == 1 and isinstance(x, _LazyConv) and x._v is None and x.act and not x.pn and not x.ups and
            F.pool_fusion_available(x.x, x.w, x.bias, x.slope)):
        # conv3d -> apply_bias -> act -> downscale3d (pgan/discriminator.py:33-44) as ONE fused op: the full-resolution
        # activation is never written (functional._ConvBiasActPool)
        try:
            return F.conv3d_act_pool(x.x, x.w, x.coef, x.bias, x.slope, x.in_info)
        except F._lib.SgError:
            pass
    x, in_info = _consume(x, premask=True)     # its gradient (an up-scale) can carry the producer's LeakyReLU mask
    return F.downscale2x(x, float(gain) / 8.0, in_info)


def upscale3d(x, factor=2):
    """networks/ops.py:276-289 (lazy: fused into the next conv3d when there is one)."""
    if factor == 1:
        return x
    if factor != 2:
        raise NotImplementedError('only factor 2 is used by the pgan path')
    return _LazyUp(x)


def upscale3d_trilinear(x, factor=2):
    """Trilinear alternative to upscale3d (half-pixel centres).  Not part of the reference's pgan (nearest only,
    ops.py:276-289): offered because BASELINE north_star names it; the networks keep the reference's upscale3d."""
    if factor == 1:
        return x
    if factor != 2:
        raise NotImplementedError('only factor 2')
    return F.upscale_trilinear2x(_val(x))


def downscale3d_trilinear(x, factor=2):
    """Trilinear x2 down-sampling with half-pixel centres samples midway between two voxels per axis: the 2x2x2 mean."""
    return avg_pool3d(x, factor)


def downscale3d(x, factor=2):
    """networks/ops.py:292-305."""
    return avg_pool3d(x, factor)


def pixel_norm(x, epsilon=1e-8):
    """networks/ops.py:308-310."""
    # fused into the conv epilogue where one wave owns all the channels of a voxel (<= 64: the sliding-halo and the
    # streamed kernel's two N tiles).  Wider layers would run the one kernel whose block owns all N tiles (conv_fwd2 with
    # NTB = 4: a single block for a whole 8^2 level, 190-410 us per launch); they keep their fast conv kernel and
    # normalise in a pass of their own, on what at those levels is a small tensor
    if isinstance(x, _LazyConv) and x._v is None and x.shape[1] <= F.PN_FUSE_MAX_CHANNELS:
        x.pn, x.eps, x.stage = True, float(epsilon), 3
        return x
    return F.pixel_norm(_val(x), float(epsilon))


def minibatch_stddev_layer(x, group_size=4):
    """networks/ops.py:313-325."""
    return F.minibatch_stddev(_val(x), group_size)


_NO_LERP_PRUNE = bool(int(os.environ.get('SARAGAN_NO_LERP_PRUNE', '0')))   # diagnostic: alpha = 0 / 1 through sg_axpby


def lerp(a, b, alpha):
    """alpha * a + (1 - alpha) * b: the fade-in of pgan/generator.py:100-101 and pgan/discriminator.py:105.
    The stabilising half of every phase runs this graph with alpha = 0 exactly (and a mixing phase starts at exactly 1):
    0 * a + 1 * b is b bit for bit (finite a) and every gradient into the faded-out branch is zero, so that branch is
    pruned here -- its value is never materialised (the lazy from_rgb / to_rgb of the previous phase does not run), and
    its variables keep the zeros their slice of the flat gradient buffer is cleared to (optimization.StepGraph._backward),
    which is what tf.gradients delivers for them."""
    dev = alpha if isinstance(alpha, F.DevCoef) else None      # captured mixing step: [alpha, 1 - alpha] on the device
    alpha = float(alpha)
    if not _NO_LERP_PRUNE and (alpha == 0.0 or alpha == 1.0):
        return _val(b) if alpha == 0.0 else _val(a)
    if dev is not None:
        return F.lerp(_val(a), _val(b), dev, F.DevCoef(1.0 - alpha, dev.buf, dev.idx + 1))
    return F.lerp(_val(a), _val(b), alpha, 1.0 - alpha)


def materialize(x):
    return _val(x)


def _unsupported(name):
    def f(*a, **kw):
        raise NotImplementedError(f'networks.ops.{name} belongs to architectures outside the pgan hot path')
    f.__name__ = name
    return f


spectral_norm = _unsupported('spectral_norm')
group_conv3d = _unsupported('group_conv3d')
instance_norm = _unsupported('instance_norm')
apply_noise = _unsupported('apply_noise')
style_mod = _unsupported('style_mod')
conv3d_depthwise = _unsupported('conv3d_depthwise')
