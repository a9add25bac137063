"""CPU oracle for the SURFGAN_3D `pgan` G+D training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU restatement (fp64 by default, fp32 on request) of the
reference's TF1 graph for the hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product (`saragan_amd/`) never does.

PARITY PINNING.  The reference ships no golden vectors or known-answer tests for this path
(SURVEY.md section 4), and its TF1 code cannot run here (no tensorflow).  The oracle is pinned by
  (1) the parameter counts the reference logged in SURFGAN_3D/out.txt:28-80 (tests/test_oracle_kat.py),
  (2) outputs of the reference's own importable PyTorch modules (pgan_pytorch/network_dict.py,
      pgan_pytorch/loss.py) run in the build container by oracle/make_golden.py and committed as
      tests/golden/ref_*.npz: Discriminator forward / input-gradient / gradient penalty,
      phase-1 Generator, EqualizedConv3d, EqualizedLinear, ChannelNormalization, Upsample, AvgPool3d,
  (3) an independent numpy loop restatement of tf.nn.conv3d (DHWIO, SAME) in oracle/conv_numpy.py.
The TF-only arithmetic (tf.train.AdamOptimizer, tf.train.ExponentialMovingAverage, the (1,2,3)-axis
GP of loss.py:140) has no runnable reference here: for those rows parity is "unpinned" and follows
the published TF1 update rules restated in SURVEY.md Appendix B.

All citations are relative to /root/reference/SURFGAN_3D unless stated otherwise.
Parameters live in a dict keyed by the TF variable names, weights in the TF
layouts: conv DHWIO [kD,kH,kW,Cin,Cout], dense [in,out].  Activations are NCDHW like the reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# bf16 emulation: the HIP path's storage rounding restated (a checker for the bf16 build; not a reference behaviour)
# ----------------------------------------------------------------------------------------------
# The bf16 build keeps activations, their gradients and the packed coef*w in bf16 and accumulates in f32.  Against the
# plain fp64 oracle that moves ~0.3 % of the LeakyReLU masks, so single gradient tensors differ by 5-20 % for a reason
# that is not a kernel error -- and a tolerance that wide cannot see one.  Inside `bf16_emulation()` this oracle rounds
# to bf16 at the points where the HIP path stores a tensor (listed at each `_q` call below, with the product's
# file:function that stores it) and takes every mask from the same values the kernels see; what is left between the two
# is accumulation order.  Gradients are rounded where they are stored as tensors (`_Q.backward`), weight gradients stay
# f32 (`_QW` is straight-through: the wgrad kernels write f32, saragan_amd/functional.py:raw_wgrad).
_EMU = {'on': False}


class bf16_emulation:
    def __enter__(self):
        self.prev = _EMU['on']
        _EMU['on'] = True
        return self

    def __exit__(self, *a):
        _EMU['on'] = self.prev


def _round_bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _Q(torch.autograd.Function):
    """A tensor stored in bf16; the gradient that arrives for it is a stored bf16 tensor too."""

    @staticmethod
    def forward(ctx, x):
        return _round_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _Q.apply(g)


class _QW(torch.autograd.Function):
    """coef*w as the packed weight image holds it (sg_conv3d_pack_weights); its gradient is the f32 wgrad output."""

    @staticmethod
    def forward(ctx, w):
        return _round_bf16(w)

    @staticmethod
    def backward(ctx, g):
        return g


def _q(x):
    return _Q.apply(x) if _EMU['on'] else x


def hip_pool_mode(n, cin, cout, d, h, w, k):
    """Which 2x2x2-pooling fusion the bf16 HIP path uses for downscale3d(leaky_relu(conv3d + b)) of this shape
    (restates saragan_amd/functional.py:_pool_mode; checked against it in tests/test_oracle.py): 1 = the conv epilogue
    stores D x W pair means and a second kernel pools H, 2 = H x W pairs then D, 3 = the epilogue stores the whole 2x2x2 mean
    (one rounding; conv_fwd3w: 32 input channels), 0 = the activation is stored and pooled."""
    if tuple(k) != (3, 3, 3) or (d | h | w) & 1 or w % 32:
        return 0
    nvox = n * d * h * w
    if cin <= 32 and cin % 8 == 0 and cout % 32 == 0 and d >= 4 and nvox >= (1 << 20):
        return 3 if cin == 32 and h >= 8 else 1
    if cin % 16 == 0 and cout % 64 == 0 and nvox >= (1 << 18):
        return 2
    return 0


def hip_subpixel(cin, cout, d, h, w, k):
    """Whether the bf16 HIP path evaluates conv3d(upscale3d(x)) of this LOW-resolution shape in sub-pixel form
    (restates saragan_amd/csrc/subpix.hip:subpix_tile and the entry point's checks; tests/test_oracle.py compares it with
    the library's answer for the shapes of the presets)."""
    if tuple(k) != (3, 3, 3) or cin % 16 or cout % 32:
        return False
    w_ = min(w, 32)
    if w_ not in (32, 16, 8):
        return False
    h_ = min(256 // w_, h, 8)
    if w_ == 32:
        h_ = min(h_, 4)
    d_ = 256 // (w_ * h_)
    if d_ < 1 or d_ > d or d_ * h_ * w_ != 256:
        return False
    return w % w_ == 0 and h % h_ == 0 and d % d_ == 0


def hip_subpixel_dgrad(cin, cout, d, h, w, k):
    """Whether the bf16 HIP path takes the gradient of conv3d(upscale3d(x)) for x in sub-pixel form (restates
    saragan_amd/csrc/subpix.hip:sg_upconv3d_subpixel_dgrad_supported for a LOW-resolution shape: whole 64-channel parts of
    gx, 16-channel chunks of gy, the forward's tiles with 32- or 16-wide rows)."""
    if not hip_subpixel(16, 32, d, h, w, k) or cin % 64 or cout % 16:
        return False
    w_ = min(w, 32)
    if w_ not in (32, 16):
        return False
    h_ = min(256 // w_, h, 8)
    if w_ == 32:
        h_ = min(h_, 4)
    return (w_, h_, 256 // (w_ * h_)) in ((32, 4, 2), (16, 8, 2))


_SUBPIX_M = ([[1., 0., 0.], [0., 1., 1.]], [[1., 1., 0.], [0., 0., 1.]])      # parity -> [tap 2][original tap 3]


def conv3d_upscaled_subpixel(x, w, activation, param=None):
    """conv3d(upscale3d(x), w) (ops.py:276-289 + :147-150) in the sub-pixel form the bf16 HIP path computes: output voxel
    2i+a of a dimension sees x[i-1] w0 + x[i] (w1 + w2) (a = 0) or x[i] (w0 + w1) + x[i+1] w2 (a = 1), so each of the 8
    parity classes is a 2x2x2 convolution of x with SUMMED weights -- which the kernel rounds to bf16 once (under
    bf16_emulation; without it this function equals the 27-tap form to rounding, tests/test_oracle.py)."""
    m = torch.tensor(_SUBPIX_M, dtype=w.dtype)
    weff = torch.einsum('aip,bjq,ckr,pqrxy->abcijkxy', m, m, m, w) * runtime_coef(w.shape, activation, param)
    if _EMU['on']:
        weff = _QW.apply(weff)
    n, _, d, h, wd = x.shape
    cout = w.shape[-1]
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    y = x.new_zeros((n, cout, 2 * d, 2 * h, 2 * wd))
    for a in range(2):
        for b in range(2):
            for c in range(2):
                # class (a, b, c): taps t read x[i + t - 1 + parity]: the window starts at padded index `parity`
                win = xp[:, :, a:a + d + 1, b:b + h + 1, c:c + wd + 1]
                y[:, :, a::2, b::2, c::2] = F.conv3d(win, weff[a, b, c].permute(4, 3, 0, 1, 2))
    return y


class _UpConvHip(torch.autograd.Function):
    """bf16 emulation of the generator's conv3d(upscale3d(x)) as the HIP path runs it: FORWARD in sub-pixel form (summed
    weights rounded once); BACKWARD: the data gradient in the same form where the library has a tile for it
    (hip_subpixel_dgrad: the transpose of the forward with the same rounded sums), else as the gather kernels compute it,
    from the 27-tap convolution with every tap's coef * w rounded on its own (functional._upconv_dgrad); the weight
    gradient does not see the weights.  First order only (nothing differentiates the generator's backward)."""

    @staticmethod
    def forward(ctx, x, w, activation, param):
        ctx.save_for_backward(x, w)
        ctx.cfg = (activation, param)
        with torch.no_grad():
            return conv3d_upscaled_subpixel(x, w, activation, param)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            x_, w_ = x.detach().requires_grad_(True), w.detach().requires_grad_(True)
            y = conv3d(upscale3d(x_), w_, *ctx.cfg)
            gx, gw = torch.autograd.grad(y, [x_, w_], gy)
            if hip_subpixel_dgrad(w.shape[3], w.shape[4], *x.shape[2:], w.shape[:3]):
                xs = x.detach().requires_grad_(True)
                (gx,) = torch.autograd.grad(conv3d_upscaled_subpixel(xs, w.detach(), *ctx.cfg), xs, gy)
        return gx, gw, None, None


def _downscale_stored(y, cin, k):
    """downscale3d of a LeakyReLU output `y` (not yet stored), with the bf16 path's rounding points."""
    if not _EMU['on'] or _TWO_D['on']:
        return downscale3d(_q(y))
    n, cout, d, h, w = y.shape
    mode = hip_pool_mode(n, cin, cout, d, h, w, k)
    if mode == 3 and _EMU.get('pool3_as_two_roundings'):      # (a deliberately displaced rounding point: make_loss_curve.py 'bf16emu2r')
        mode = 1
    if mode == 1:
        return _q(F.avg_pool3d(_q(F.avg_pool3d(y, (2, 1, 2))), (1, 2, 1)))
    if mode == 2:
        return _q(F.avg_pool3d(_q(F.avg_pool3d(y, (1, 2, 2))), (2, 1, 1)))
    if mode == 3:
        return _q(F.avg_pool3d(y, 2))
    return _q(downscale3d(_q(y)))


# ----------------------------------------------------------------------------------------------
# networks/ops.py
# ----------------------------------------------------------------------------------------------
def k_rule(x: int) -> int:
    """networks/ops.py:25-29 legacy per-dimension kernel rule."""
    return 1 if x < 3 else 3


def calculate_gain(activation: str, param=None) -> float:
    """networks/ops.py:60-77."""
    linear_fns = ['linear', 'conv1d', 'conv2d', 'conv3d', 'conv_transpose1d', 'conv_transpose2d',
                  'conv_transpose3d']
    if activation in linear_fns or activation == 'sigmoid':
        return 1.0
    if activation == 'tanh':
        return 5.0 / 3
    if activation == 'relu':
        return math.sqrt(2.0)
    if activation == 'leaky_relu':
        assert param is not None
        if (not isinstance(param, bool) and isinstance(param, int)) or isinstance(param, float):
            return math.sqrt(2.0 / (1 + param ** 2))
        raise ValueError("negative_slope {} not a valid number".format(param))
    raise ValueError("Unsupported nonlinearity {}".format(activation))


def runtime_coef(shape: Sequence[int], activation: str, param=None, lrmul: float = 1.0) -> float:
    """networks/ops.py:111-116: he_std * lrmul with fan_in = prod(shape[:-1])."""
    fan_in = float(np.prod(shape[:-1]))
    return calculate_gain(activation, param) / math.sqrt(fan_in) * lrmul


def conv3d(x: torch.Tensor, w: torch.Tensor, activation: str, param=None) -> torch.Tensor:
    """networks/ops.py:147-150.  w: raw variable, DHWIO.  tf.nn.conv3d stride 1 'SAME' NCDHW is a
    cross-correlation with zero padding k//2 per side for odd k."""
    kd, kh, kw = w.shape[:3]
    assert kd % 2 == 1 and kh % 2 == 1 and kw % 2 == 1, "SAME == k//2 only for odd kernels"
    wt = w * runtime_coef(w.shape, activation, param)
    if _EMU['on']:
        wt = _QW.apply(wt)
    wt = wt.permute(4, 3, 0, 1, 2)
    return F.conv3d(x, wt, padding=(kd // 2, kh // 2, kw // 2))


def dense(x: torch.Tensor, w: torch.Tensor, activation: str, param=None) -> torch.Tensor:
    """networks/ops.py:139-144.  Flatten is C-major of NCDHW (tf.reshape of an NCDHW tensor)."""
    if x.dim() > 2:
        x = x.reshape(x.shape[0], -1)
    wt = w * runtime_coef(w.shape, activation, param)
    if _EMU['on'] and w.shape[1] != 1:      # the one-unit logit layer runs in f32 (saragan_amd/networks/ops.py:dense)
        wt = _QW.apply(wt)
    return x @ wt


def apply_bias(x: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """networks/ops.py:130-136 (lrmul = 1)."""
    if x.dim() == 2:
        return x + b
    return x + b.reshape(1, -1, 1, 1, 1)


class _LeakyReluRef(torch.autograd.Function):
    """networks/ops.py:167-182: y = max(x, a x); dx = where(y >= 0, dy, a dy) (subgradient 1 at 0),
    and the same mask again for the second-order term."""

    @staticmethod
    def forward(ctx, x, a):
        y = torch.maximum(x, x * a)
        ctx.save_for_backward(y)
        ctx.a = a
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _LeakyReluMask.apply(dy, y, ctx.a), None


class _LeakyReluMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, y, a):
        ctx.save_for_backward(y)
        ctx.a = a
        return torch.where(y >= 0, dy, dy * a)

    @staticmethod
    def backward(ctx, ddx):
        (y,) = ctx.saved_tensors
        return _LeakyReluMask.apply(ddx, y, ctx.a), None, None


def leaky_relu(x: torch.Tensor, alpha_lr: float = 0.2) -> torch.Tensor:
    return _LeakyReluRef.apply(x, alpha_lr)


def act(x: torch.Tensor, activation: str, param=None) -> torch.Tensor:
    """networks/ops.py:185-192."""
    if activation == 'leaky_relu':
        assert param is not None
        return leaky_relu(x, param)
    if activation == 'linear':
        return x
    raise ValueError(f"Unknown activation {activation}")


def pixel_norm(x: torch.Tensor, epsilon: float = 1e-8) -> torch.Tensor:
    """networks/ops.py:308-310."""
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + epsilon)


_TWO_D = {'on': False}      # set by the loss builders for the 2-D tree (SURFGAN_2D/networks/ops.py:176-231)


def upscale3d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """networks/ops.py:250-262,276-289: nearest-neighbour x2 (tile + batch_to_space); the custom
    gradient (8 * avg-pool = sum-pool) is what autograd derives for repeat_interleave.  2-D tree: upscale2d (H, W only)."""
    if factor == 1:
        return x
    if _TWO_D['on']:
        return x.repeat_interleave(factor, 3).repeat_interleave(factor, 4)
    return x.repeat_interleave(factor, 2).repeat_interleave(factor, 3).repeat_interleave(factor, 4)


def downscale3d(x: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """networks/ops.py:265-273,292-305: avg_pool3d k=s=2 VALID.  2-D tree: downscale2d (avg_pool2d)."""
    if factor == 1:
        return x
    if _TWO_D['on']:
        return F.avg_pool3d(x, (1, factor, factor))
    return F.avg_pool3d(x, factor)


def minibatch_stddev_layer(x: torch.Tensor, group_size: int = 4) -> torch.Tensor:
    """networks/ops.py:313-325 (disabled in pgan: pgan/discriminator.py:50)."""
    g = min(group_size, x.shape[0])
    s = x.shape
    y = x.reshape(g, -1, s[1], s[2], s[3], s[4])
    y = y - y.mean(dim=0, keepdim=True)
    y = (y * y).mean(dim=0)
    y = torch.sqrt(y + 1e-8)
    y = y.mean(dim=[1, 2, 3, 4], keepdim=True)
    y = y.repeat(g, 1, s[2], s[3], s[4])
    return torch.cat([x, y], dim=1)


def alpha_update(alpha: float, mixing_nimg: int, starting_alpha: float, batch_size: int,
                 global_size: int) -> float:
    """networks/ops.py:4-23 (fp32 variable arithmetic)."""
    if mixing_nimg == 0:
        return 0.0
    num_steps = mixing_nimg // (batch_size * global_size)
    upd = np.float32(starting_alpha / num_steps)
    return float(max(np.float32(alpha) - upd, np.float32(0)))


def num_filters(phase: int, base_shape: Sequence[int], size: str) -> int:
    """networks/ops.py:201-236."""
    lists = {
        'xxs': [256, 256, 64, 32, 16, 8, 4, 2], 'xs': [256, 256, 64, 64, 32, 16, 8, 4],
        's': [512, 512, 128, 128, 64, 32, 16, 8], 'm': [1024, 1024, 256, 256, 128, 64, 32, 16],
        'l': [2048, 2048, 512, 512, 256, 128, 64, 32],
        'xl': [4096, 4096, 1024, 1024, 512, 256, 128, 64],
        'xxl': [8192, 8192, 2048, 1024, 1024, 512, 256, 128]}
    if size not in lists:
        raise ValueError(f"Unknown size: {size}")
    current_dim = [2 ** (phase - 1) * d for d in base_shape[1:]]
    log_product = np.log2(np.prod(current_dim))
    reference_log = [4 + n * 3 for n in range(0, 7)]
    index = int(np.argmin(np.abs(np.array(reference_log) - log_product)))
    return lists[size][index]


def preset_specs(size: str, base_shape: Sequence[int], num_phases: int):
    """filter_spec / kernel_spec that reproduce the legacy presets (SURVEY section 8 header):
    fs[l-1] = [F_l, F_l], ks[l-1] = [k, k] with k per dim = 1 if dim < 3 else 3 (ops.py:25-29)."""
    fs, ks = [], []
    for l in range(1, num_phases + 1):
        f = num_filters(l, base_shape, size)
        fs.append([f, f])
        dims = [d * 2 ** (l - 1) for d in base_shape[1:]]
        kk = [k_rule(d) for d in dims]
        ks.append([kk, kk])
    return ks, fs


# ----------------------------------------------------------------------------------------------
# networks/pgan/{generator,discriminator}.py
# ----------------------------------------------------------------------------------------------
def _spec(spec, phase_i, layer_i):
    """pgan/generator.py:4-24 / pgan/discriminator.py:3-23: ValueError on missing entries."""
    if phase_i >= len(spec):
        raise ValueError
    if layer_i >= len(spec[phase_i]):
        raise ValueError
    return spec[phase_i][layer_i]


def variable_shapes(phase: int, base_shape: Sequence[int], latent_dim: int, kernel_spec, filter_spec
                    ) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of the trainable variables the reference creates at `phase`
    (pgan/generator.py:79-98, pgan/discriminator.py:76-107, ops.py:118,131)."""
    ch = base_shape[0]
    v0 = int(np.prod(base_shape[1:]))
    fs, ks = filter_spec, kernel_spec
    out: Dict[str, Tuple[int, ...]] = {}
    g = 'generator/'
    out[g + 'generator_in/dense/weight'] = (latent_dim, v0 * _spec(fs, 0, 0))
    out[g + 'generator_in/dense/bias'] = (v0 * _spec(fs, 0, 0),)
    out[g + 'generator_in/conv/weight'] = (*_spec(ks, 0, 1), _spec(fs, 0, 0), _spec(fs, 0, 1))
    out[g + 'generator_in/conv/bias'] = (_spec(fs, 0, 1),)
    c_prev = _spec(fs, 0, 1)
    for i in range(2, phase + 1):
        if i == phase:
            out[g + f'to_rgb_{phase - 1}/weight'] = (1, 1, 1, c_prev, ch)
            out[g + f'to_rgb_{phase - 1}/bias'] = (ch,)
        out[g + f'generator_block_{i}/conv_1/weight'] = (*_spec(ks, i - 1, 0), c_prev, _spec(fs, i - 1, 0))
        out[g + f'generator_block_{i}/conv_1/bias'] = (_spec(fs, i - 1, 0),)
        out[g + f'generator_block_{i}/conv_2/weight'] = (*_spec(ks, i - 1, 1), _spec(fs, i - 1, 0),
                                                        _spec(fs, i - 1, 1))
        out[g + f'generator_block_{i}/conv_2/bias'] = (_spec(fs, i - 1, 1),)
        c_prev = _spec(fs, i - 1, 1)
    out[g + f'to_rgb_{phase}/weight'] = (1, 1, 1, c_prev, ch)
    out[g + f'to_rgb_{phase}/bias'] = (ch,)

    d = 'discriminator/'
    out[d + f'from_rgb_{phase}/weight'] = (1, 1, 1, ch, _spec(fs, phase - 1, 1))
    out[d + f'from_rgb_{phase}/bias'] = (_spec(fs, phase - 1, 1),)
    c_in = _spec(fs, phase - 1, 1)
    for i in reversed(range(2, phase + 1)):
        out[d + f'discriminator_block_{i}/conv_1/weight'] = (*_spec(ks, i - 1, 1), c_in, _spec(fs, i - 1, 0))
        out[d + f'discriminator_block_{i}/conv_1/bias'] = (_spec(fs, i - 1, 0),)
        out[d + f'discriminator_block_{i}/conv_2/weight'] = (*_spec(ks, i - 1, 0), _spec(fs, i - 1, 0),
                                                            _spec(fs, i - 2, 1))
        out[d + f'discriminator_block_{i}/conv_2/bias'] = (_spec(fs, i - 2, 1),)
        c_in = _spec(fs, i - 2, 1)
        if i == phase:
            out[d + f'from_rgb_{phase - 1}/weight'] = (1, 1, 1, ch, _spec(fs, phase - 2, 1))
            out[d + f'from_rgb_{phase - 1}/bias'] = (_spec(fs, phase - 2, 1),)
    out[d + 'discriminator_out/weight'] = (*_spec(ks, 0, 1), c_in, _spec(fs, 0, 0))
    out[d + 'discriminator_out/bias'] = (_spec(fs, 0, 0),)
    out[d + 'discriminator_out/dense_1/weight'] = (v0 * _spec(fs, 0, 0), latent_dim)
    out[d + 'discriminator_out/dense_1/bias'] = (latent_dim,)
    out[d + 'discriminator_out/dense_2/weight'] = (latent_dim, 1)
    out[d + 'discriminator_out/dense_2/bias'] = (1,)
    return out


def init_params(phase, base_shape, latent_dim, kernel_spec, filter_spec, seed=0,
                dtype=torch.float64, bias_std: float = 0.0, arch: str = 'pgan') -> Params:
    """weight ~ N(0,1) (ops.py:118-119, lrmul 1), bias zeros (ops.py:131).  `bias_std` > 0 draws
    non-zero biases so that parity tests exercise the bias path."""
    gen = torch.Generator().manual_seed(seed)
    p: Params = {}
    shapes = ARCHS[arch][2](phase, base_shape, latent_dim, kernel_spec, filter_spec)
    for name, shape in shapes.items():
        if name.endswith('weight'):
            p[name] = torch.randn(shape, generator=gen, dtype=torch.float64).to(dtype)
        else:
            p[name] = (torch.randn(shape, generator=gen, dtype=torch.float64) * bias_std).to(dtype)
    return p


def generator(p: Params, z, alpha, phase, base_shape, activation, kernel_spec, filter_spec,
              param=None, conditioning=None, torch_port_order=False):
    """pgan/generator.py:74-103 (+ generator_in :26-45, generator_block :48-71).
    torch_port_order: the second stage of every block as the reference's OWN PyTorch port orders it,
    conv -> bias -> norm -> act (pgan_pytorch/network_dict.py:287-289) where the TF graph has conv -> bias -> act -> norm
    (pgan/generator.py:66-70).  Only used to pin this restatement against fixtures produced by running that port
    (tests/golden/ref_generator_p{2,3}.npz); the product mirrors the TF graph."""
    if conditioning is not None:
        raise NotImplementedError()
    g = 'generator/'
    fs = filter_spec
    # _q: the tensors the bf16 HIP path stores -- each fused conv + bias + LeakyReLU + pixel-norm launch writes its result
    # once (saragan_amd/networks/ops.py:_LazyConv.value); above 64 channels pixel_norm is a pass of its own
    # (networks/ops.py:pixel_norm, functional.PN_FUSE_MAX_CHANNELS) and the activation is stored in between
    def pn_stored(a):
        return _q(pixel_norm(a if a.shape[1] <= 64 else _q(a)))

    x = dense(_q(z), p[g + 'generator_in/dense/weight'], activation, param)
    x = _q(act(apply_bias(x, p[g + 'generator_in/dense/bias']), activation, param))
    x = x.reshape(-1, _spec(fs, 0, 0), *base_shape[1:])
    x = conv3d(x, p[g + 'generator_in/conv/weight'], activation, param)
    x = pn_stored(act(apply_bias(x, p[g + 'generator_in/conv/bias']), activation, param))
    x_upsample = None
    for i in range(2, phase + 1):
        if i == phase and not (_EMU['on'] and float(alpha) == 0.0):     # (pruned at alpha = 0: networks/ops.py:lerp)
            t = conv3d(x, p[g + f'to_rgb_{phase - 1}/weight'], 'linear')
            x_upsample = upscale3d(_q(apply_bias(t, p[g + f'to_rgb_{phase - 1}/bias'])))
        b = g + f'generator_block_{i}/'
        w1 = p[b + 'conv_1/weight']
        if _EMU['on'] and not _TWO_D['on'] and hip_subpixel(x.shape[1], w1.shape[-1], *x.shape[2:], w1.shape[:3]):
            x = _UpConvHip.apply(x, w1, activation, param)      # (forward: the summed weights are what gets rounded)
        else:
            x = conv3d(upscale3d(x), w1, activation, param)
        x = pn_stored(act(apply_bias(x, p[b + 'conv_1/bias']), activation, param))
        x = conv3d(x, p[b + 'conv_2/weight'], activation, param)
        if torch_port_order:
            x = act(pixel_norm(apply_bias(x, p[b + 'conv_2/bias'])), activation, param)
        else:
            x = pn_stored(act(apply_bias(x, p[b + 'conv_2/bias']), activation, param))
    x_out = _q(apply_bias(conv3d(x, p[g + f'to_rgb_{phase}/weight'], 'linear'), p[g + f'to_rgb_{phase}/bias']))
    if x_upsample is not None:
        if _EMU['on'] and float(alpha) == 1.0:
            return x_upsample
        x_out = _q(alpha * x_upsample + (1 - alpha) * x_out)
    return x_out


def discriminator(p: Params, x, alpha, phase, latent_dim, activation, kernel_spec, filter_spec,
                  param=None, conditioning=None):
    """pgan/discriminator.py:71-108 (+ discriminator_block :25-45, discriminator_out :48-68)."""
    if conditioning is not None:
        raise NotImplementedError()
    d = 'discriminator/'
    x_downscale = x
    x = conv3d(x, p[d + f'from_rgb_{phase}/weight'], activation, param)
    x = _q(act(apply_bias(x, p[d + f'from_rgb_{phase}/bias']), activation, param))
    for i in reversed(range(2, phase + 1)):
        b = d + f'discriminator_block_{i}/'
        x = conv3d(x, p[b + 'conv_1/weight'], activation, param)
        x = _q(act(apply_bias(x, p[b + 'conv_1/bias']), activation, param))
        cin = x.shape[1]
        x = conv3d(x, p[b + 'conv_2/weight'], activation, param)
        x = act(apply_bias(x, p[b + 'conv_2/bias']), activation, param)
        x = _downscale_stored(x, cin, p[b + 'conv_2/weight'].shape[:3])     # (the fused tail: functional._ConvBiasActPool)
        if i == phase:
            if _EMU['on'] and float(alpha) == 0.0:      # the faded-out branch is pruned (networks/ops.py:lerp)
                continue
            t = conv3d(_q(downscale3d(x_downscale)), p[d + f'from_rgb_{phase - 1}/weight'], activation, param)
            t = _q(act(apply_bias(t, p[d + f'from_rgb_{phase - 1}/bias']), activation, param))
            x = t if (_EMU['on'] and float(alpha) == 1.0) else _q(alpha * t + (1 - alpha) * x)
    x = conv3d(x, p[d + 'discriminator_out/weight'], activation, param)
    x = _q(act(apply_bias(x, p[d + 'discriminator_out/bias']), activation, param))
    x = dense(x, p[d + 'discriminator_out/dense_1/weight'], activation, param)
    x = _q(act(apply_bias(x, p[d + 'discriminator_out/dense_1/bias']), activation, param))
    x = dense(x, p[d + 'discriminator_out/dense_2/weight'], 'linear')      # f32 logits (networks/ops.py:dense)
    return apply_bias(x, p[d + 'discriminator_out/dense_2/bias'])


# ----------------------------------------------------------------------------------------------
# networks/pgandeep/{generator,discriminator}.py: N = len(kernel_spec[phase]) convolutions per block
# ----------------------------------------------------------------------------------------------
def variable_shapes_deep(phase: int, base_shape: Sequence[int], latent_dim: int, kernel_spec, filter_spec
                         ) -> Dict[str, Tuple[int, ...]]:
    """Variables of pgandeep in creation order (pgandeep/generator.py:26-122, pgandeep/discriminator.py:25-131).
    tf.get_variable sizes a conv weight from the tensor that reaches it (ops.py:148): track the running width."""
    ch = base_shape[0]
    v0 = int(np.prod(base_shape[1:]))
    fs, ks = filter_spec, kernel_spec
    out: Dict[str, Tuple[int, ...]] = {}
    g = 'generator/'
    c = _spec(fs, 0, 0)
    out[g + 'generator_in/dense/weight'] = (latent_dim, v0 * c)
    out[g + 'generator_in/dense/bias'] = (v0 * c,)
    for j in range(1, len(ks[0])):                                   # pgandeep/generator.py:37-43
        out[g + f'generator_in/conv_{j}/weight'] = (*_spec(ks, 0, j), c, _spec(fs, 0, j))
        out[g + f'generator_in/conv_{j}/bias'] = (_spec(fs, 0, j),)
        c = _spec(fs, 0, j)
    for i in range(2, phase + 1):
        if i == phase:
            out[g + f'to_rgb_{phase - 1}/weight'] = (1, 1, 1, c, ch)
            out[g + f'to_rgb_{phase - 1}/bias'] = (ch,)
        for j in range(1, len(ks[i - 1]) + 1):                       # pgandeep/generator.py:63-71
            out[g + f'generator_block_{i}/conv_{j}/weight'] = (*_spec(ks, i - 1, j - 1), c, _spec(fs, i - 1, j - 1))
            out[g + f'generator_block_{i}/conv_{j}/bias'] = (_spec(fs, i - 1, j - 1),)
            c = _spec(fs, i - 1, j - 1)
    out[g + f'to_rgb_{phase}/weight'] = (1, 1, 1, c, ch)
    out[g + f'to_rgb_{phase}/bias'] = (ch,)
    d = 'discriminator/'
    c = _spec(fs, phase - 1, 1)                                      # pgandeep/discriminator.py:112
    out[d + f'from_rgb_{phase}/weight'] = (1, 1, 1, ch, c)
    out[d + f'from_rgb_{phase}/bias'] = (c,)
    for i in reversed(range(2, phase + 1)):
        n = len(ks[i - 1])
        for j in range(1, n + 1):                                    # pgandeep/discriminator.py:27-38
            nf = _spec(fs, i - 2, n - 1) if j == n else _spec(fs, i - 1, n - j - 1)
            out[d + f'discriminator_block_{i}/conv_{j}/weight'] = (*_spec(ks, i - 1, 1), c, nf)
            out[d + f'discriminator_block_{i}/conv_{j}/bias'] = (nf,)
            c = nf
        if i == phase:
            out[d + f'from_rgb_{phase - 1}/weight'] = (1, 1, 1, ch, _spec(fs, phase - 2, 1))
            out[d + f'from_rgb_{phase - 1}/bias'] = (_spec(fs, phase - 2, 1),)
    n0 = len(ks[0])
    for j in range(1, n0):                                           # pgandeep/discriminator.py:66-72
        nf = _spec(fs, 0, n0 - j - 1)
        out[d + f'discriminator_out/conv_{j}/weight'] = (*_spec(ks, 0, n0 - j), c, nf)
        out[d + f'discriminator_out/conv_{j}/bias'] = (nf,)
        c = nf
    out[d + 'discriminator_out/dense_1/weight'] = (v0 * c, latent_dim)
    out[d + 'discriminator_out/dense_1/bias'] = (latent_dim,)
    out[d + 'discriminator_out/dense_2/weight'] = (latent_dim, 1)
    out[d + 'discriminator_out/dense_2/bias'] = (1,)
    return out


def generator_deep(p: Params, z, alpha, phase, base_shape, activation, kernel_spec, filter_spec, param=None,
                   conditioning=None):
    """pgandeep/generator.py:97-122."""
    if conditioning is not None:
        raise NotImplementedError()
    g = 'generator/'

    def stage(x, scope):
        x = conv3d(x, p[scope + '/weight'], activation, param)
        return pixel_norm(act(apply_bias(x, p[scope + '/bias']), activation, param))

    x = dense(z, p[g + 'generator_in/dense/weight'], activation, param)
    x = act(apply_bias(x, p[g + 'generator_in/dense/bias']), activation, param)
    x = x.reshape(-1, _spec(filter_spec, 0, 0), *base_shape[1:])
    for j in range(1, len(kernel_spec[0])):
        x = stage(x, g + f'generator_in/conv_{j}')
    x_upsample = None
    for i in range(2, phase + 1):
        if i == phase:
            t = conv3d(x, p[g + f'to_rgb_{phase - 1}/weight'], 'linear')
            x_upsample = upscale3d(apply_bias(t, p[g + f'to_rgb_{phase - 1}/bias']))
        x = upscale3d(x)
        for j in range(1, len(kernel_spec[i - 1]) + 1):
            x = stage(x, g + f'generator_block_{i}/conv_{j}')
    x_out = apply_bias(conv3d(x, p[g + f'to_rgb_{phase}/weight'], 'linear'), p[g + f'to_rgb_{phase}/bias'])
    if x_upsample is not None:
        x_out = alpha * x_upsample + (1 - alpha) * x_out
    return x_out


def discriminator_deep(p: Params, x, alpha, phase, latent_dim, activation, kernel_spec, filter_spec, param=None,
                       conditioning=None):
    """pgandeep/discriminator.py:97-131."""
    if conditioning is not None:
        raise NotImplementedError()
    d = 'discriminator/'

    def stage(x, scope):
        x = conv3d(x, p[scope + '/weight'], activation, param)
        return act(apply_bias(x, p[scope + '/bias']), activation, param)

    x_downscale = x
    x = stage(x, d + f'from_rgb_{phase}')
    for i in reversed(range(2, phase + 1)):
        for j in range(1, len(kernel_spec[i - 1]) + 1):
            x = stage(x, d + f'discriminator_block_{i}/conv_{j}')
        x = downscale3d(x)
        if i == phase:
            t = stage(downscale3d(x_downscale), d + f'from_rgb_{phase - 1}')
            x = alpha * t + (1 - alpha) * x
    for j in range(1, len(kernel_spec[0])):
        x = stage(x, d + f'discriminator_out/conv_{j}')
    x = dense(x, p[d + 'discriminator_out/dense_1/weight'], activation, param)
    x = act(apply_bias(x, p[d + 'discriminator_out/dense_1/bias']), activation, param)
    x = dense(x, p[d + 'discriminator_out/dense_2/weight'], 'linear')
    return apply_bias(x, p[d + 'discriminator_out/dense_2/bias'])


ARCHS = {'pgan': (generator, discriminator, variable_shapes),
         'pgandeep': (generator_deep, discriminator_deep, variable_shapes_deep)}


# ----------------------------------------------------------------------------------------------
# networks/loss.py  (all randomness is injected: TF and torch RNG streams cannot be matched)
# ----------------------------------------------------------------------------------------------
def _softplus(x):
    return F.softplus(x)


def forward_simultaneous(p: Params, real, z, noise_real, noise_fake, gamma, alpha, phase, base_shape,
                         latent_dim, kernel_spec, filter_spec, activation, leakiness, loss_fn,
                         gp_weight, noise_stddev, arch='pgan', two_d=False):
    """networks/loss.py:101-165, including quirk Q1: slopes reduce over axes (1,2,3) of the 5-D
    gradient, so slopes has shape [N, W] (loss.py:140).  two_d: the 2-D tree on D == 1 volumes -- its version of this
    function reduces the 4-D gradient over (1,2,3) = every non-batch axis (SURFGAN_2D/networks/loss.py:130-137), here
    axes (1,2,3,4) with slopes [N] -> [N,1], and its up/down-sampling leaves D alone (upscale2d / downscale2d)."""
    net = dict(phase=phase, activation=activation, kernel_spec=kernel_spec, filter_spec=filter_spec,
               param=leakiness)
    generator, discriminator = ARCHS[arch][:2]      # networks.<arch> (optuna_objective.py:64-65)
    _TWO_D['on'] = bool(two_d)
    gen_sample = generator(p, z, alpha, base_shape=base_shape, **net)
    if _EMU['on']:
        # saragan_amd/networks/loss.py: images and noise enter in bf16, sg_axpby adds in f32 and stores bf16; the
        # interpolates are ONE kernel since round 4 (sg_lerp_rows: f32 weights, f32 arithmetic, one rounding); with the wgan loss
        # D(real) and D(fake) are ONE pass over the concatenated batch (forward_simultaneous, `link`), which is what decides the
        # pooling fusion per layer
        real_n = _q(_q(real) + _q(noise_real) * noise_stddev)
        fake_n = _q(gen_sample + _q(noise_fake) * noise_stddev)
        interpolates = _q(gamma * real_n + (1 - gamma) * fake_n.detach()).detach().requires_grad_(True)
        if loss_fn == 'wgan':
            both = discriminator(p, torch.cat([real_n, fake_n], dim=0), alpha, latent_dim=latent_dim, **net)
            disc_real, disc_fake_g = both[:real.shape[0]], both[real.shape[0]:]
            disc_fake_d = disc_fake_g
        else:
            disc_fake_g = discriminator(p, fake_n, alpha, latent_dim=latent_dim, **net)
            disc_fake_d = disc_fake_g
            disc_real = discriminator(p, real_n, alpha, latent_dim=latent_dim, **net)
    else:
        real_n = real + noise_real * noise_stddev
        fake_n = gen_sample + noise_fake * noise_stddev
        disc_fake_d = discriminator(p, fake_n.detach(), alpha, latent_dim=latent_dim, **net)
        disc_real = discriminator(p, real_n, alpha, latent_dim=latent_dim, **net)
        interpolates = (gamma * real_n + (1 - gamma) * fake_n.detach()).requires_grad_(True)
    d_int = discriminator(p, interpolates, alpha, latent_dim=latent_dim, **net)
    (gradients,) = torch.autograd.grad(d_int.sum(), interpolates, create_graph=True)
    gradients = _q(gradients)       # (emulation: from_rgb's data gradient is a stored bf16 tensor)
    if two_d:
        slopes = torch.sqrt(torch.sum(gradients * gradients, dim=(1, 2, 3, 4))).reshape(-1, 1)
    else:
        slopes = torch.sqrt(torch.sum(gradients * gradients, dim=(1, 2, 3)))
    if not _EMU['on']:
        disc_fake_g = discriminator(p, fake_n, alpha, latent_dim=latent_dim, **net)
    if loss_fn == 'wgan':
        gp_loss = gp_weight * (slopes - 1) ** 2
        disc_loss = disc_fake_d - disc_real
        drift_loss = 1e-3 * disc_real ** 2
        disc_loss = torch.mean(disc_loss + gp_loss + drift_loss)
        gen_loss = -torch.mean(disc_fake_g)
    elif loss_fn == 'logistic':
        gp_loss = gp_weight * torch.mean(slopes ** 2)
        disc_loss = torch.mean(_softplus(disc_fake_d)) + torch.mean(_softplus(-disc_real))
        disc_loss = disc_loss + gp_loss
        gen_loss = torch.mean(_softplus(-disc_fake_g))
    else:
        raise ValueError(f"Unknown loss function: {loss_fn}")
    _TWO_D['on'] = False
    return gen_loss, disc_loss, gp_loss, gen_sample


def forward_discriminator(p: Params, real, z, noise_real, noise_fake, gamma, alpha, phase, base_shape,
                         latent_dim, kernel_spec, filter_spec, activation, leakiness, loss_fn,
                         gp_weight, noise_stddev, arch='pgan'):
    """networks/loss.py:42-98 (alternate mode; GP reduces over (1,2,3,4): loss.py:79)."""
    net = dict(phase=phase, activation=activation, kernel_spec=kernel_spec, filter_spec=filter_spec,
               param=leakiness)
    generator, discriminator = ARCHS[arch][:2]      # networks.<arch> (optuna_objective.py:64-65)
    gen_sample = generator(p, z, alpha, base_shape=base_shape, **net)
    real_n = real + noise_real * noise_stddev
    fake_n = gen_sample + noise_fake * noise_stddev
    disc_fake_d = discriminator(p, fake_n.detach(), alpha, latent_dim=latent_dim, **net)
    disc_real = discriminator(p, real_n, alpha, latent_dim=latent_dim, **net)
    interpolates = (gamma * real_n + (1 - gamma) * fake_n.detach()).requires_grad_(True)
    d_int = discriminator(p, interpolates, alpha, latent_dim=latent_dim, **net)
    (gradients,) = torch.autograd.grad(d_int.sum(), interpolates, create_graph=True)
    slopes = torch.sqrt(torch.sum(gradients * gradients, dim=(1, 2, 3, 4)))
    if loss_fn == 'wgan':
        gp_loss = gp_weight * (slopes - 1) ** 2
        # [N,1] + [N] broadcasts to [N,N] in TF exactly as it does here (loss.py:82-86).
        disc_loss = torch.mean((disc_fake_d - disc_real) + gp_loss + 1e-3 * disc_real ** 2)
    elif loss_fn == 'logistic':
        gp_loss = gp_weight * torch.mean(slopes ** 2)
        disc_loss = torch.mean(_softplus(disc_fake_d)) + torch.mean(_softplus(-disc_real)) + gp_loss
    else:
        raise ValueError(f"Unknown loss function: {loss_fn}")
    return disc_loss, gp_loss


def forward_generator(p: Params, real, z, noise_real, noise_fake, alpha, phase, base_shape, latent_dim,
                      kernel_spec, filter_spec, activation, leakiness, loss_fn, noise_stddev, arch='pgan'):
    """networks/loss.py:4-39."""
    net = dict(phase=phase, activation=activation, kernel_spec=kernel_spec, filter_spec=filter_spec,
               param=leakiness)
    generator, discriminator = ARCHS[arch][:2]      # networks.<arch> (optuna_objective.py:64-65)
    gen_sample = generator(p, z, alpha, base_shape=base_shape, **net)
    fake_n = gen_sample + noise_fake * noise_stddev
    disc_fake_g = discriminator(p, fake_n, alpha, latent_dim=latent_dim, **net)
    if loss_fn == 'wgan':
        gen_loss = -torch.mean(disc_fake_g)
    elif loss_fn == 'logistic':
        gen_loss = torch.mean(_softplus(-disc_fake_g))
    else:
        raise ValueError(f"Unknown loss function: {loss_fn}")
    return gen_sample, gen_loss


# ----------------------------------------------------------------------------------------------
# optimization.py / ExtendedEMA.py / third-party update rules
# ----------------------------------------------------------------------------------------------
class TFAdam:
    """tf.train.AdamOptimizer(lr, beta1, beta2, epsilon=1e-8) as called at optimization.py:16,28:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, beta1=0.0, beta2=0.9, epsilon=1e-8):
        self.b1, self.b2, self.eps = beta1, beta2, epsilon
        self.t = 0
        self.m: Params = {}
        self.v: Params = {}

    def apply(self, params: Params, grads: Dict[str, torch.Tensor], lr: float):
        self.t += 1
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for name, g in grads.items():
            if name not in self.m:
                self.m[name] = torch.zeros_like(params[name])
                self.v[name] = torch.zeros_like(params[name])
            self.m[name] = self.b1 * self.m[name] + (1 - self.b1) * g
            self.v[name] = self.b2 * self.v[name] + (1 - self.b2) * g * g
            params[name] = params[name] - lr_t * self.m[name] / (torch.sqrt(self.v[name]) + self.eps)


class TFSGD:
    """tf.train.GradientDescentOptimizer (optimization.py:17-18): theta -= lr * g."""

    def __init__(self):
        self.t = 0

    def apply(self, params: Params, grads: Dict[str, torch.Tensor], lr: float):
        self.t += 1
        for name, g in grads.items():
            params[name] = params[name] - lr * g


class TFMomentum:
    """tf.train.MomentumOptimizer(lr, momentum, use_nesterov) (optimization.py:21-22 passes use_nesterov=True);
    TF training_ops ApplyMomentum: accum = accum * momentum + g;
    var -= g * lr + accum * momentum * lr (Nesterov)  |  var -= accum * lr."""

    def __init__(self, momentum=0.9, use_nesterov=True):
        self.mom, self.nesterov = momentum, use_nesterov
        self.t = 0
        self.accum: Params = {}

    def apply(self, params: Params, grads: Dict[str, torch.Tensor], lr: float):
        self.t += 1
        for name, g in grads.items():
            if name not in self.accum:
                self.accum[name] = torch.zeros_like(params[name])
            self.accum[name] = self.accum[name] * self.mom + g
            if self.nesterov:
                params[name] = params[name] - (g * lr + self.accum[name] * self.mom * lr)
            else:
                params[name] = params[name] - self.accum[name] * lr


class TFAdadelta:
    """tf.train.AdadeltaOptimizer(lr, rho, epsilon=1e-07) (optimization.py:19-20); TF training_ops ApplyAdadelta:
    accum = accum * rho + g^2 * (1 - rho); update = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
    var -= update * lr; accum_update = accum_update * rho + update^2 * (1 - rho)."""

    def __init__(self, rho=0.95, epsilon=1e-7):
        self.rho, self.eps = rho, epsilon
        self.t = 0
        self.accum: Params = {}
        self.accum_update: Params = {}

    def apply(self, params: Params, grads: Dict[str, torch.Tensor], lr: float):
        self.t += 1
        for name, g in grads.items():
            if name not in self.accum:
                self.accum[name] = torch.zeros_like(params[name])
                self.accum_update[name] = torch.zeros_like(params[name])
            self.accum[name] = self.accum[name] * self.rho + g * g * (1 - self.rho)
            update = torch.sqrt(self.accum_update[name] + self.eps) * torch.rsqrt(self.accum[name] + self.eps) * g
            params[name] = params[name] - update * lr
            self.accum_update[name] = self.accum_update[name] * self.rho + update * update * (1 - self.rho)


def clip_by_global_norm(grads: Dict[str, torch.Tensor], clip_norm: float = 1.0):
    """tf.clip_by_global_norm (optimization.py:66-67): g * clip / max(global_norm, clip)."""
    gn = torch.sqrt(sum((g * g).sum() for g in grads.values()))
    scale = clip_norm / max(float(gn), clip_norm)
    return {k: g * scale for k, g in grads.items()}, gn


def ema_update(shadow: Params, params: Params, decay: float):
    """tf.train.ExponentialMovingAverage without num_updates / zero_debias (ExtendedEMA.py:11-19):
    shadow -= (1-decay) * (shadow - theta)."""
    for k in params:
        shadow[k] = shadow[k] - (1 - decay) * (shadow[k] - params[k])


def lr_update(intra_phase_step: int, steps_per_phase: int, lr_max: float, lr_increase, lr_decrease,
              lr_rise_niter, lr_decay_niter) -> float:
    """optimization.py:227-296.  Returns the fp32 value assigned to the lr variable."""
    lr = np.float32(lr_max)
    if lr_increase or lr_decrease:
        a = np.float32(lr_max / 100)
        if lr_increase == 'linear':
            if intra_phase_step < lr_rise_niter:
                lr = np.float32(intra_phase_step / lr_rise_niter) * np.float32(lr_max)
        elif lr_increase == 'exponential':
            if intra_phase_step < lr_rise_niter:
                b_rise = np.float32(np.log(100) / lr_rise_niter)
                lr = a * np.exp(b_rise * np.float32(intra_phase_step), dtype=np.float32)
        if lr_decrease:
            step_decay_start = steps_per_phase - lr_decay_niter
            remaining = steps_per_phase - intra_phase_step
            if lr_decrease == 'linear':
                if intra_phase_step > step_decay_start:
                    lr = np.float32(remaining / lr_decay_niter) * np.float32(lr_max)
            elif lr_decrease == 'exponential':
                if intra_phase_step > step_decay_start:
                    b_decay = np.float32(np.log(100) / lr_decay_niter)
                    lr = a * np.exp(b_decay * np.float32(remaining), dtype=np.float32)
    return float(lr)


def split_vars(p: Params):
    gen = [k for k in p if k.startswith('generator/')]
    disc = [k for k in p if k.startswith('discriminator/')]
    return gen, disc


def step_simultaneous(p: Params, adam_g: TFAdam, adam_d: TFAdam, shadow: Optional[Params], rnd: dict,
                      real, alpha, cfg: dict, g_lr: float, d_lr: float, freeze: Optional[Sequence[str]] = None,
                      ema_beta: float = 0.99, g_clipping=False, d_clipping=False, world_grads=None):
    """One `simultaneous` optimisation step (optimization.py:128-163,
    optuna_objective.py:463-467): both gradients at the pre-step weights, freeze = names that are
    NOT updated (the previous phase's variables while mixing, Q4).  Returns the fetches and grads."""
    work = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    gen_loss, disc_loss, gp_loss, gen_sample = forward_simultaneous(
        work, real, rnd['z'], rnd['noise_real'], rnd['noise_fake'], rnd['gamma'], alpha, **cfg)
    gnames, dnames = split_vars(work)
    if freeze is not None:
        fz = set(freeze)
        gnames = [k for k in gnames if k not in fz]
        dnames = [k for k in dnames if k not in fz]
    # (allow_unused: under bf16_emulation the faded-out branch of alpha = 0 is pruned as in the product; tf.gradients
    # delivers zeros for its variables either way)
    g_grads = torch.autograd.grad(gen_loss, [work[k] for k in gnames], retain_graph=True, allow_unused=_EMU['on'])
    d_grads = torch.autograd.grad(disc_loss, [work[k] for k in dnames], allow_unused=_EMU['on'])
    g_grads = {k: (g.detach() if g is not None else torch.zeros_like(work[k])) for k, g in zip(gnames, g_grads)}
    d_grads = {k: (g.detach() if g is not None else torch.zeros_like(work[k])) for k, g in zip(dnames, d_grads)}
    if g_clipping:
        g_grads, _ = clip_by_global_norm(g_grads)
    if d_clipping:
        d_grads, _ = clip_by_global_norm(d_grads)
    adam_g.apply(p, g_grads, g_lr)
    adam_d.apply(p, d_grads, d_lr)
    if shadow is not None:
        ema_update(shadow, p, ema_beta)
    return dict(gen_loss=gen_loss.detach(), disc_loss=disc_loss.detach(), gp_loss=gp_loss.detach(),
                gen_sample=gen_sample.detach(), g_grads=g_grads, d_grads=d_grads)


def step_alternate(p: Params, adam_g: TFAdam, adam_d: TFAdam, shadow: Optional[Params], rnd: dict,
                   real, alpha, cfg: dict, g_lr: float, d_lr: float, freeze: Optional[Sequence[str]] = None,
                   ema_beta: float = 0.99):
    """One `alternate` optimisation step (optimization.py:165-220): the discriminator is updated from
    forward_discriminator's loss first; under tf.control_dependencies([train_disc]) the generator loss is then built
    on the UPDATED discriminator and the generator is updated.  Both loss builders draw their own randomness in the
    reference; with injected randomness they are fed the same z / noise tensors (as the product's InjectedRandom does)."""
    gnames, dnames = split_vars(p)
    if freeze is not None:
        fz = set(freeze)
        gnames = [k for k in gnames if k not in fz]
        dnames = [k for k in dnames if k not in fz]
    gcfg = {k: v for k, v in cfg.items() if k != 'gp_weight'}
    work = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    disc_loss, gp_loss = forward_discriminator(work, real, rnd['z'], rnd['noise_real'], rnd['noise_fake'], rnd['gamma'],
                                               alpha, **cfg)
    d_grads = torch.autograd.grad(disc_loss, [work[k] for k in dnames])
    d_grads = dict(zip(dnames, [g.detach() for g in d_grads]))
    adam_d.apply(p, d_grads, d_lr)
    work = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    gen_sample, gen_loss = forward_generator(work, real, rnd['z'], rnd['noise_real'], rnd['noise_fake'], alpha, **gcfg)
    g_grads = torch.autograd.grad(gen_loss, [work[k] for k in gnames])
    g_grads = dict(zip(gnames, [g.detach() for g in g_grads]))
    adam_g.apply(p, g_grads, g_lr)
    if shadow is not None:
        ema_update(shadow, p, ema_beta)
    return dict(gen_loss=gen_loss.detach(), disc_loss=disc_loss.detach(), gp_loss=gp_loss.detach(),
                gen_sample=gen_sample.detach(), g_grads=g_grads, d_grads=d_grads)


def specs_2d(num_phases: int, size: str):
    """The 2-D pgan (SURFGAN_2D/networks/pgan/*.py, legacy signature with base_dim = num_filters(1)) written as
    kernel / filter specs of the 3-D restatement on D == 1 volumes: base_shape (C,1,4,4), kernels (1,3,3) (k() of the
    2-D extents, all >= 3), filters num_filters_2d(l) (SURFGAN_2D/networks/ops.py:139-158)."""
    lists = {
        'xxs': [64] * 8 + [32, 16, 8, 4, 2], 'xs': [128] * 8 + [64, 32, 16, 8, 4], 's': [256] * 8 + [128, 64, 32, 16, 8],
        'm': [512] * 8 + [256, 128, 64, 32, 16], 'l': [512] * 9 + [256, 128, 64, 32],
        'xl': [1024] * 9 + [512, 256, 128, 64], 'xxl': [2048] * 9 + [1024, 512, 256, 128]}
    fl = lists[size][-num_phases:]
    fs = [[f, f] for f in fl]
    ks = [[[1, 3, 3], [1, 3, 3]] for _ in fl]
    return ks, fs


def draw_randomness(n, latent_dim, img_shape, seed, dtype=torch.float64):
    """The four random tensors of loss.py:116-133, drawn from a seeded torch generator."""
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64).to(dtype)
    return dict(z=r(n, latent_dim), noise_real=r(n, *img_shape), noise_fake=r(n, *img_shape),
                gamma=torch.rand(n, 1, 1, 1, 1, generator=gen, dtype=torch.float64).to(dtype))
