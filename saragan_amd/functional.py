"""Differentiable wrappers over the C ABI (include/saragan_hip.h): every op below launches a hand-written
gfx950 kernel of libsaragan_hip.so on the current torch stream; torch only owns the device buffers and the
autograd tape.  There is NO fallback: tensors must live on the GPU.

Tensors keep the reference's logical NCDHW shape (SURFGAN_3D/networks/ops.py:150,273) but are stored
channels-last (torch.channels_last_3d == NDHWC); 2-D tensors [N, F] are NDHWC with one voxel per sample.
Each backward is itself built from these Functions, so second-order gradients (the gradient penalty of
networks/loss.py:133-140 differentiates through D's data gradient) work.
"""
import contextlib
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ConvEpilogue, ConvShape, check

_DT = {torch.float32: _lib.SG_F32, torch.bfloat16: _lib.SG_BF16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f'saragan_amd supports float32 and bfloat16 activations, got {t.dtype}')


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('saragan_amd ops run on the GPU only (no CPU fallback); got a CPU tensor')


def ndhwc(x):
    """Returns x stored as NDHWC (5-D: channels_last_3d; 2-D: row-major)."""
    if x.dim() == 5:
        return x.contiguous(memory_format=torch.channels_last_3d)
    if x.dim() == 2:
        return x.contiguous()
    raise ValueError(f'expected a [N,C,D,H,W] or [N,F] tensor, got shape {tuple(x.shape)}')


def _dims(x):
    """-> (n, c, d, h, w) of an NCDHW / [N,F] tensor."""
    if x.dim() == 5:
        n, c, d, h, w = x.shape
        return n, c, d, h, w
    n, c = x.shape
    return n, c, 1, 1, 1


def _empty_like_shape(x, c_out, spatial=None, dtype=None):
    n, _, d, h, w = _dims(x)
    if spatial is not None:
        d, h, w = spatial
    dtype = dtype or x.dtype
    if x.dim() == 5:
        return torch.empty((n, c_out, d, h, w), device=x.device, dtype=dtype, memory_format=torch.channels_last_3d)
    return torch.empty((n, c_out), device=x.device, dtype=dtype)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# ---------------------------------------------------------------------------------------------------
# raw launches (no autograd)
# ---------------------------------------------------------------------------------------------------
def _shape(n, d, h, w, cin, cout, k, ups=False):
    return ConvShape(n, d, h, w, cin, cout, k[0], k[1], k[2], 1 if ups else 0)


_PACK_CACHE = {}
_PACK_STATE = {'stale': False}
_NO_PACK_BATCH = bool(int(os.environ.get('SARAGAN_NO_PACK_BATCH', '0')))   # diagnostic: drop the images, one pack launch per layer and use
PACK_STATS = {'single': 0, 'batches': 0, 'batched': 0}      # counters for the tests (host side only)


def clear_pack_cache():
    """Packed-weight images are reused while the parameter is unchanged (same storage, same version counter): the
    four discriminator passes and their gradients all read one image per orientation."""
    _PACK_CACHE.clear()
    _SUBPIX_CACHE.clear()
    _PACK_STATE['stale'] = False


def mark_packs_stale():
    """The parameters were rewritten behind torch's version counters (the optimiser kernels; a graph capture, which records
    launches without running them).  The fragment images stay where they are and are rewritten together, one launch for all
    layers (sg_conv3d_pack_weights_batch), when the next convolution asks for one -- 44 pack launches per step at the
    benchmarked configuration otherwise.  The few images of other kinds (sub-pixel forms, to_rgb matrices) are dropped and
    rebuilt on use."""
    if _NO_PACK_BATCH:
        return clear_pack_cache()
    for k in [k for k in _PACK_CACHE if k[0] == 'rgbmat']:
        del _PACK_CACHE[k]
    _SUBPIX_CACHE.clear()
    _PACK_STATE['stale'] = bool(_PACK_CACHE)


def live_packs():
    """The cached images and the kept filter-gradient workspaces (tensors): a captured graph whose launches carry their
    addresses keeps them alive -- the caches themselves are released when the next step graph is built."""
    return [v[0] for v in _PACK_CACHE.values()] + list(_CLEAN_WS.values())


def _refresh_packs(lib, st):
    _PACK_STATE['stale'] = False
    by_dt = {}
    for key in list(_PACK_CACHE):
        if key[0] == 'rgbmat':
            continue
        wp, w, shp, w32 = _PACK_CACHE[key]
        if w._version != key[1] or w.data_ptr() != key[0]:      # modified through torch since: a new key is in use, this one is dead
            del _PACK_CACHE[key]
            continue
        by_dt.setdefault(key[4], []).append((w32, key[2], 1 if key[3] else 0, wp, shp))
    for dt, items in by_dt.items():
        n = len(items)
        ws = (C.c_void_p * n)(*[it[0].data_ptr() for it in items])
        coefs = (C.c_float * n)(*[it[1] for it in items])
        flips = (C.c_int * n)(*[it[2] for it in items])
        wps = (C.c_void_p * n)(*[it[3].data_ptr() for it in items])
        shapes = (ConvShape * n)(*[it[4] for it in items])
        check(lib.sg_conv3d_pack_weights_batch(n, ws, coefs, flips, wps, shapes, dt, st), 'sg_conv3d_pack_weights_batch')
        PACK_STATS['batches'] += 1
        PACK_STATS['batched'] += n


def _packed(w, coef, flip, shp, dt, lib, st):
    if _PACK_STATE['stale']:
        _refresh_packs(lib, st)
    key = (w.data_ptr(), w._version, float(coef), bool(flip), dt, shp.cin, shp.cout, shp.kd, shp.kh, shp.kw)
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit[0]
    w32 = w.detach()
    if w32.dtype != torch.float32 or not w32.is_contiguous():
        w32 = w32.contiguous().float()
    wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=w.device, dtype=torch.uint8)
    check(lib.sg_conv3d_pack_weights(_ptr(w32), float(coef), 1 if flip else 0, _ptr(wp), C.byref(shp), dt, st),
          'sg_conv3d_pack_weights')
    PACK_STATS['single'] += 1
    # only long-lived tensors (parameters) are cached, not transient gradients -- and only f32 contiguous ones are refreshed
    # in place (w32 aliases the parameter: the batch reads the values of the day)
    # (a VIEW of a parameter -- the 2-D networks hand [kh, kw, ci, co] filters over as [1, kh, kw, ci, co] -- lives as long as the
    # parameter and shares its version counter: configs[4] packed its 42 filters 126 times per step before this was recognised)
    base = w._base if w._is_view() else None
    long_lived = w.is_leaf or not w.requires_grad or (base is not None and base.is_leaf)
    if long_lived and w32.data_ptr() == w.data_ptr():
        keep = ConvShape(shp.n, shp.d, shp.h, shp.w, shp.cin, shp.cout, shp.kd, shp.kh, shp.kw, shp.upsample_in)
        # holding the tensor keeps its storage (and so the key) from being recycled; for a view its DETACHED alias is held (same
        # storage, same version counter, no autograd node of the step that made the view)
        _PACK_CACHE[key] = (wp, w if (w.is_leaf or not w.requires_grad) else w32, keep, w32)
    return wp


def _empty_signs(device, n, d, h, w, c):
    """Sign words of an NDHWC tensor (include/saragan_hip.h: sg_sign_words): int32 [n,d,h,w,ceil(c/32)]."""
    return torch.empty((n, d, h, w, (c + 31) // 32), device=device, dtype=torch.int32)


def _check_signs(bits, nvox, c):
    if bits is None:
        return
    if bits.dtype != torch.int32 or not bits.is_contiguous() or bits.numel() != nvox * ((c + 31) // 32):
        raise ValueError('sign words must be a contiguous int32 tensor [n,d,h,w,ceil(c/32)] of the masked tensor')


def sign_words(t):
    """Sign words of t (sg_sign_words): what sg_conv_epilogue.sign_out writes for a conv output."""
    lib = _lib.load()
    _req_cuda(t)
    t = ndhwc(t)
    n, c, d, h, w = _dims(t)
    out = _empty_signs(t.device, n, d, h, w, c)
    check(lib.sg_sign_words(_ptr(t), _ptr(out), n * d * h * w, c, _dt(t), _stream()), 'sg_sign_words')
    return out


# Sub-pixel form of upscale3d -> conv3d (sg_upconv3d_subpixel_fwd: one launch for all eight parity classes, 3.4x fewer
# MFMAs than the 27-tap fused gather).  SARAGAN_NO_SUBPIXEL=1 keeps the gather kernels (A/B, tests).  (Round 2 had this form
# as eight launches of the streamed kernel: 1.6 % slower per step than the gather, the phases too short to cover their halo
# DMA and the stride-2 scatter writing half lines.)
_NO_SUBPIXEL = bool(int(os.environ.get('SARAGAN_NO_SUBPIXEL', '0')))
_SUBPIX_CACHE = {}


def _subpixel_packed(w, coef, shp, dt, lib, st):
    """The summed weights of the eight classes (sg_upconv3d_subpixel_pack), cached like the other packed images."""
    key = (w.data_ptr(), w._version, float(coef), dt)
    hit = _SUBPIX_CACHE.get(key)
    if hit is not None:
        return hit[0]
    w32 = w.detach()
    if w32.dtype != torch.float32 or not w32.is_contiguous():
        w32 = w32.contiguous().float()
    wp = torch.empty(lib.sg_upconv3d_subpixel_packed_bytes(C.byref(shp), dt), device=w.device, dtype=torch.uint8)
    check(lib.sg_upconv3d_subpixel_pack(_ptr(w32), float(coef), _ptr(wp), C.byref(shp), dt, st), 'sg_upconv3d_subpixel_pack')
    if w.is_leaf or not w.requires_grad:
        _SUBPIX_CACHE[key] = (wp, w)
    return wp


def _raw_upconv_subpixel(x, w, coef, bias, act, slope, pixel_norm, eps, want_scale, want_signs):
    """y = epilogue(conv3d(upscale3d(x), coef*w)) on the low-resolution input, all eight parity classes in one launch.
    Returns None when the library has no sub-pixel kernel for the request (caller runs the fused-gather convolution)."""
    lib = _lib.load()
    n, cin, d, h, wd = _dims(x)
    cout = w.shape[-1]
    dt = _dt(x)
    if dt != _lib.SG_BF16 or x.dim() != 5:
        return None
    st = _stream()
    shp = _shape(n, d, h, wd, cin, cout, (3, 3, 3), False)
    if not lib.sg_upconv3d_subpixel_packed_bytes(C.byref(shp), dt):
        return None
    y = _empty_like_shape(x, cout, (2 * d, 2 * h, 2 * wd))
    signs = _empty_signs(x.device, n, 2 * d, 2 * h, 2 * wd, cout) if want_signs else None
    scale = torch.empty(n * 8 * d * h * wd, device=x.device, dtype=torch.float32) if (pixel_norm and want_scale) else None
    b32 = bias.detach().contiguous().float() if bias is not None else None
    ep = ConvEpilogue(_ptr(b32), 1 if act else 0, float(slope), 1 if pixel_norm else 0, float(eps), _ptr(scale), None, 0.0,
                      _ptr(signs))
    # (the weights are packed only once the library has accepted the shape: a probe launch costs nothing it would not do anyway)
    wp = _subpixel_packed(w, coef, shp, dt, lib, st)
    rc = lib.sg_upconv3d_subpixel_fwd(_ptr(x), _ptr(wp), _ptr(y), C.byref(shp), C.byref(ep), dt, st)
    if rc == _lib.SG_EUNSUPPORTED:
        return None
    check(rc, 'sg_upconv3d_subpixel_fwd')
    return y, scale, signs


def raw_conv(x, w, coef, flip, ups=False, bias=None, act=False, slope=0.2, pixel_norm=False, eps=1e-8,
             want_scale=False, mask_bits=None, mask_slope=0.0, want_signs=False, pool=False, pn_bwd=None, rgb=None, pw_bwd=None):
    """y = epilogue(conv3d(x, coef*w)) with w in DHWIO; `flip` selects the data-gradient weights.
    Returns (y, pixel-norm scale or None, sign words of y or None).  pool (sg_conv_epilogue.pool): 1 -- y is the
    2 x 1 x 2 (D x H x W) block mean of the output, [n,cout,d/2,h,w/2]; 2 -- the 1 x 2 x 2 block mean, [n,cout,d,h/2,w/2];
    3 -- the whole 2 x 2 x 2 block mean [n,cout,d/2,h/2,w/2] (32 input channels);
    returns None if no kernel of the build fuses it here.  pn_bwd = (y, scale) of a pixel-norm stage: the result is
    pushed through that stage's backward in the epilogue (sg_conv_epilogue.pn_bwd_y; mask_bits = the stage's sign words);
    None if no kernel does that for this layer.
    rgb = (matrix [cout] f32, bias tensor or None): to_rgb (one image channel) of the stored output in the epilogue
    (sg_conv_epilogue.rgb_*); returns (y, scale, signs, img), or None if the library has no such epilogue for this layer.
    pw_bwd = dict(x=image, wmat=[cout] f32, want_dx, dw, db, coef): from_rgb's whole backward in the epilogue of this data
    gradient (sg_conv_epilogue.pw_*; y is not written); returns the image gradient (or True when none was asked for), or None
    if the library declines."""
    lib = _lib.load()
    _req_cuda(x, w, bias)
    x = ndhwc(x)
    n, cx, d, h, wd = _dims(x)
    if w.dim() == 2:
        w = w.reshape(1, 1, 1, *w.shape)
    kd, kh, kw, wi, wo = w.shape
    cin, cout = (wo, wi) if flip else (wi, wo)
    if cx != cin:
        raise ValueError(f'conv3d: input has {cx} channels, weight expects {cin}')
    if (ups and not flip and (kd, kh, kw) == (3, 3, 3) and not _NO_SUBPIXEL and mask_bits is None and not pool and
            pn_bwd is None):
        res = _raw_upconv_subpixel(x, w, coef, bias, act, slope, pixel_norm, eps, want_scale, want_signs)
        if res is not None:
            return res
    if ups:
        d, h, wd = 2 * d, 2 * h, 2 * wd
    shp = _shape(n, d, h, wd, cin, cout, (kd, kh, kw), ups)
    dt = _dt(x)
    st = _stream()
    wp = _packed(w, coef, flip, shp, dt, lib, st)
    pool = int(pool)
    if pool and ({1: d | wd, 2: h | wd, 3: d | h | wd}[pool] & 1 or x.dim() != 5):
        return None
    y = None if pw_bwd is not None else \
        _empty_like_shape(x, cout, {0: (d, h, wd), 1: (d // 2, h, wd // 2), 2: (d, h // 2, wd // 2), 3: (d // 2, h // 2, wd // 2)}[pool])
    _check_signs(mask_bits, n * d * h * wd, cout)
    signs = _empty_signs(x.device, n, d, h, wd, cout) if want_signs else None
    scale = None
    if pixel_norm and want_scale:
        scale = torch.empty(n * d * h * wd, device=x.device, dtype=torch.float32)
    b32 = bias.detach().contiguous().float() if bias is not None else None
    ep = ConvEpilogue(_ptr(b32), 1 if act else 0, float(slope), 1 if pixel_norm else 0, float(eps), _ptr(scale),
                      _ptr(mask_bits), float(mask_slope), _ptr(signs))
    ep.pool = pool
    if pn_bwd is not None:
        pn_y, pn_scale = ndhwc(pn_bwd[0]), pn_bwd[1]
        if tuple(pn_y.shape) != tuple(y.shape) or pn_y.dtype != y.dtype or pn_scale.numel() != n * d * h * wd:
            raise ValueError('pn_bwd: y / scale do not match the convolution output')
        ep.pn_bwd_y, ep.pn_bwd_scale = pn_y.data_ptr(), pn_scale.data_ptr()
    ws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
    if ws_bytes and not pool:      # scratch for the library's two-pass (K-split) path of this layer
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        ep.workspace, ep.workspace_bytes = ws.data_ptr(), ws_bytes
    img = dimg = None
    if rgb is not None:
        img = _empty_like_shape(x, 1, (d, h, wd))
        ep.rgb_w, ep.rgb_bias, ep.rgb_out = rgb[0].data_ptr(), (rgb[1].data_ptr() if rgb[1] is not None else None), img.data_ptr()
    if pw_bwd is not None:
        pws = lib.sg_conv3d_pw_epilogue_workspace()
        ws = torch.empty(pws, device=x.device, dtype=torch.uint8)
        ep.workspace, ep.workspace_bytes = ws.data_ptr(), pws
        dimg = _empty_like_shape(x, 1, (d, h, wd)) if pw_bwd['want_dx'] else None
        ep.pw_x, ep.pw_wmat, ep.pw_dx = pw_bwd['x'].data_ptr(), pw_bwd['wmat'].data_ptr(), (dimg.data_ptr() if dimg is not None else None)
        ep.pw_dw = pw_bwd['dw'].data_ptr() if pw_bwd['dw'] is not None else None
        ep.pw_dbias = pw_bwd['db'].data_ptr() if pw_bwd['db'] is not None else None
        ep.pw_coef = float(pw_bwd['coef'])
    rc = lib.sg_conv3d_fwd(_ptr(x), _ptr(wp), _ptr(y), C.byref(shp), C.byref(ep), dt, st)
    if (pool or pn_bwd is not None or rgb is not None or pw_bwd is not None) and rc == _lib.SG_EUNSUPPORTED:
        return None
    check(rc, 'sg_conv3d_fwd')
    if pw_bwd is not None:
        return dimg if dimg is not None else True
    if rgb is not None:
        return y, scale, signs, img
    return y, scale, signs


_CLEAN_WS = {}            # (device, workspace bytes) -> kept workspace, zero between calls (SG_WGRAD_CLEAN_WORKSPACE)
_CLEAN_WS_DECLINED = set()   # shapes whose kernels do not leave the workspace clean (pointwise / small-channel paths)
_NO_CLEAN_WS = bool(int(os.environ.get('SARAGAN_NO_CLEAN_WS', '0')))   # diagnostic: a fresh workspace + memset per weight gradient


def clear_kept_workspaces():
    """Releases the kept filter-gradient workspaces (a new phase has other layer shapes)."""
    _CLEAN_WS.clear()
    _CLEAN_WS_DECLINED.clear()


def _wgrad_launch(lib, x, dy, mask, mask_slope, gain, dw, db, coef, accumulate, shp, dt):
    """One filter (+ bias) gradient through sg_conv3d_wgrad_bias_ex.  The kernels sum their tiles into a zeroed workspace; the
    workspace of a layer is kept between calls and cleared by the pass that reads it (SG_WGRAD_CLEAN_WORKSPACE) instead of a
    hipMemsetAsync per call -- 37 launches per step at the benchmarked configuration.  Returns the library's code."""
    flags = _lib.SG_WGRAD_ACCUMULATE if accumulate else 0
    ws_bytes = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
    skey = (shp.n, shp.d, shp.h, shp.w, shp.cin, shp.cout, shp.kd, shp.kh, shp.kw, shp.upsample_in, dt, mask is not None)
    if not _NO_CLEAN_WS and skey not in _CLEAN_WS_DECLINED:
        key = (x.device, ws_bytes)
        ws = _CLEAN_WS.get(key)
        if ws is None:
            ws = _CLEAN_WS[key] = torch.zeros(ws_bytes, device=x.device, dtype=torch.uint8)
        rc = lib.sg_conv3d_wgrad_bias_ex(_ptr(x), _ptr(dy), _ptr(mask), float(mask_slope), float(gain), _ptr(dw), _ptr(db), float(coef),
                                         flags | _lib.SG_WGRAD_CLEAN_WORKSPACE, _ptr(ws), ws_bytes, C.byref(shp), dt, _stream())
        if rc == 0:
            return 0
        if rc != _lib.SG_EUNSUPPORTED:
            _CLEAN_WS.pop(key, None)         # a failed call leaves the workspace undefined
            return rc
        _CLEAN_WS_DECLINED.add(skey)         # (declined before anything was touched)
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    return lib.sg_conv3d_wgrad_bias_ex(_ptr(x), _ptr(dy), _ptr(mask), float(mask_slope), float(gain), _ptr(dw), _ptr(db), float(coef),
                                       flags, _ptr(ws), ws_bytes, C.byref(shp), dt, _stream())


def raw_wgrad(x, dy, k, coef, ups=False, want_db=False, w_ptr=0, b_ptr=0):
    """dw[kd,kh,kw,cin,cout] (f32) = coef * sum_v x[v+tap] (x) dy[v]; optionally db[cout] = sum_v dy[v].
    w_ptr / b_ptr: data_ptr of the parameters these are the gradients of (grads_into)."""
    lib = _lib.load()
    _req_cuda(x, dy)
    x, dy = ndhwc(x), ndhwc(dy)
    if x.dtype != dy.dtype:
        raise TypeError('wgrad: x and dy dtypes differ')
    n, cin, _, _, _ = _dims(x)
    n2, cout, d, h, w = _dims(dy)
    dt = _dt(x)
    if ups and tuple(k) == (3, 3, 3) and not _NO_SUBPIXEL and x.dim() == 5:
        # conv3d(upscale3d(x)): 64 (class, tap) tiles on the low-resolution grid instead of 8 x 27 tap products per voxel
        low = _shape(n, d // 2, h // 2, w // 2, cin, cout, (3, 3, 3), False)
        if lib.sg_upconv3d_subpixel_wgrad_supported(C.byref(low), dt):
            ws_bytes = lib.sg_upconv3d_subpixel_wgrad_workspace(C.byref(low), dt)
            ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
            dw = _f32_out(w_ptr, (3, 3, 3, cin, cout), x.device)
            db = _f32_out(b_ptr, (cout,), x.device) if want_db else None
            check(lib.sg_upconv3d_subpixel_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), float(coef), _ptr(ws), ws_bytes,
                                                 C.byref(low), dt, _stream()), 'sg_upconv3d_subpixel_wgrad')
            return dw, db
    shp = _shape(n, d, h, w, cin, cout, k, ups)
    acc = _grad_acc(w_ptr, (k[0], k[1], k[2], cin, cout))
    if acc is not None:       # a further contribution to a gradient that is in its slot already: added there, nothing returned
        db = _f32_out(b_ptr, (cout,), x.device) if want_db else None      # (the bias gradient is written, not added)
        rc = _wgrad_launch(lib, x, dy, None, 0.0, 1.0, acc, db, coef, True, shp, dt)
        if rc != _lib.SG_EUNSUPPORTED:
            check(rc, 'sg_conv3d_wgrad_bias_ex (accumulate)')
            _note_accumulated(w_ptr)
            return None, db
        _unclaim(b_ptr, db)
    dw = _f32_out(w_ptr, (k[0], k[1], k[2], cin, cout), x.device)
    db = _f32_out(b_ptr, (cout,), x.device) if want_db else None
    check(_wgrad_launch(lib, x, dy, None, 0.0, 1.0, dw, db, coef, False, shp, dt), 'sg_conv3d_wgrad_bias_ex')
    return dw, db


def raw_bias_act_bwd(dy, y, slope, want_dx=True, want_db=False, b_ptr=0):
    lib = _lib.load()
    dy = ndhwc(dy)
    n, c, d, h, w = _dims(dy)
    nvox = n * d * h * w
    bits = y is not None and y.dtype == torch.int32     # the mask as sign words instead of the activation itself
    if bits:
        _check_signs(y, nvox, c)
    elif y is not None:
        y = ndhwc(y)
    dx = torch.empty_like(dy) if want_dx else None
    db = ws = None
    if want_db:
        db = _f32_out(b_ptr, (c,), dy.device)
        ws = torch.empty(lib.sg_bias_act_bwd_workspace(c), device=dy.device, dtype=torch.uint8)
    fn = lib.sg_bias_act_bwd_bits if bits else lib.sg_bias_act_bwd
    check(fn(_ptr(dy), _ptr(y), _ptr(dx), _ptr(db), _ptr(ws), nvox, c, float(slope), _dt(dy), _stream()),
          'sg_bias_act_bwd')
    return dx, db


# ---------------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------------
_NO_SIGN_WORDS = bool(int(os.environ.get('SARAGAN_NO_SIGN_WORDS', '0')))   # diagnostic: activation-based masks only
_ZERO = {}


def _zero_scalar(like):
    """A shared 0-d placeholder for 'no bias gradient' outputs (a fresh new_zeros(()) is one fill launch each,
    ~60 per step)."""
    z = _ZERO.get(like.device)
    if z is None:
        z = _ZERO[like.device] = torch.zeros((), device=like.device, dtype=torch.float32)
    return z.detach()    # a new tensor object over the same storage: autograd may tag it per Function


_SKIP = {'ptrs': frozenset()}
_GRAD_DEST = {}      # parameter data_ptr -> [f32 view of the step's flat gradient buffer, claimed]
_NO_GRAD_DEST = bool(int(os.environ.get('SARAGAN_NO_GRAD_DEST', '0')))   # diagnostic: gradients as tensors of their own, added by autograd
ACCUMULATED_IN_PLACE = set()   # data_ptr of the parameters whose slot took a later contribution in place during this backward
GRAD_DEST_STATS = {'claimed': 0, 'accumulated': 0, 'adopted': 0, 'copied': 0, 'unreached': 0}   # counters for the tests (host side only)


@contextlib.contextmanager
def grads_into(dest):
    """dest: {parameter data_ptr: view of the flat gradient buffer shaped like the parameter}.  Inside this context (the
    torch.autograd.backward of one training step, optimization.StepGraph._backward) the FIRST weight / bias gradient a
    backward Function computes for a registered parameter is written by its kernel straight into that view and handed to
    autograd as such: AccumulateGrad keeps a gradient tensor nobody else holds as .grad, so the parameter's gradient is
    in the optimiser's buffer without the zero-fill + `add` launch per parameter (71 per step at the benchmarked
    configuration, 0.37 ms).  Further contributions to the same parameter (the gradient penalty's second-order term of the
    discriminator's weights) are tensors of their own, summed by autograd; a sum that ended up outside the buffer is
    copied in by the caller (StepGraph._land)."""
    prev = dict(_GRAD_DEST)
    _GRAD_DEST.clear()
    ACCUMULATED_IN_PLACE.clear()
    if not _NO_GRAD_DEST:
        _GRAD_DEST.update({k: [v, False] for k, v in dest.items()})
    try:
        yield
    finally:
        _GRAD_DEST.clear()
        _GRAD_DEST.update(prev)


def _grad_out(ptr, shape):
    """A fresh alias (f32, `shape`) of the flat-buffer slot registered for parameter `ptr` if no gradient has been written
    to it in this backward, else None.  Never while a graph of the backward is being recorded (create_graph)."""
    if not ptr or not _GRAD_DEST or torch.is_grad_enabled():
        return None
    ent = _GRAD_DEST.get(ptr)
    if ent is None or ent[1] or ent[0].numel() != math.prod(shape):
        return None
    ent[1] = True
    GRAD_DEST_STATS['claimed'] += 1
    return ent[0].view(shape)


def _grad_acc(ptr, shape):
    """The slot registered for parameter `ptr` when a gradient HAS been written to it in this backward (by a kernel of this
    module: the alias is with autograd, waiting for the parameter's other contributions) -- a kernel that accumulates
    (sg_conv3d_wgrad_bias_ex, SG_WGRAD_ACCUMULATE) adds the next contribution in place and the Function returns None for it."""
    if not ptr or not _GRAD_DEST or torch.is_grad_enabled():
        return None
    ent = _GRAD_DEST.get(ptr)
    if ent is None or not ent[1] or ent[0].numel() != math.prod(shape):
        return None
    return ent[0].view(shape)


def _note_accumulated(ptr):
    """A kernel HAS added a contribution in place to the slot of parameter `ptr` (the launch was not declined).  From here on the
    slot holds more than what autograd was handed: sound only while every later contribution is made in place too -- one that
    arrives as a tensor of its own makes the engine sum OUT of place (V1 + T2 elsewhere), and copying that sum over the slot would
    drop what was added here.  StepGraph._land refuses that case."""
    ACCUMULATED_IN_PLACE.add(ptr)
    GRAD_DEST_STATS['accumulated'] += 1


def _unclaim(ptr, t):
    """The launch that was to write `t` declined (SG_EUNSUPPORTED: nothing was written) and the caller takes another path: if
    `t` is the slot of parameter `ptr`, the slot is free again.  (A slot that stayed claimed without being handed to autograd
    would take later contributions in place -- _grad_acc -- while the parameter's .grad is built from other tensors.)"""
    ent = _GRAD_DEST.get(ptr) if (ptr and t is not None) else None
    if ent is not None and ent[1] and t.data_ptr() == ent[0].data_ptr():
        ent[1] = False
        GRAD_DEST_STATS['claimed'] -= 1


def _f32_out(ptr, shape, device):
    out = _grad_out(ptr, shape)
    return out if out is not None else torch.empty(shape, device=device, dtype=torch.float32)


@contextlib.contextmanager
def skip_param_grads(params):
    """Inside this context the backward Functions do not compute gradients of the given parameters.
    `ctx.needs_input_grad` is True for every weight that requires grad, also when the caller asked autograd for other
    inputs only (the gradient-penalty d D/d x pass, networks/loss.py:136-139; the generator loss flowing through the
    discriminator, optimization.py:128-163): the weight/bias-gradient kernels would run and their results be dropped."""
    prev = _SKIP['ptrs']
    _SKIP['ptrs'] = prev | frozenset(p.data_ptr() for p in params)
    try:
        yield
    finally:
        _SKIP['ptrs'] = prev


def _wants(ctx, i, ptr):
    return ctx.needs_input_grad[i] and ptr not in _SKIP['ptrs']


class ActInfo:
    """Bookkeeping of one LeakyReLU output `a` (created by the layer that produced it).  Consumers that are able
    to apply the LeakyReLU-backward mask where(a >= 0, g, slope*g) inside THEIR backward kernel (the data-gradient
    conv's epilogue, the masked up-scale) register as `premask`; if every consumer did, the producer skips its own
    mask pass and takes the bias gradient from the weight-gradient kernel.  Evaluated at backward time, when all
    consumers are known."""

    def __init__(self, slope):
        self.slope = float(slope)
        self.bits = None          # sign words of `a`, written by the producing conv's epilogue
        self.pn = None            # (y, scale) when the stage goes on through pixel_norm: y = pixel_norm(a); consumers that
        #                           registered as premask then apply the pixel-norm backward as well (_dgrad_into)
        self.pw = None            # from_rgb (1x1x1 from one image channel): dict(x, w, coef, w_ptr, b_ptr, has_b, x_req) -- a consumer
        #                           whose data-gradient kernel has the pw_* epilogue runs this layer's whole backward there ...
        self.pw_result = None     # ... and leaves (gx, gw, gb) here for the layer's own backward to return (_dgrad_into)
        self.n_consumers = 0
        self.n_premask = 0

    def consume(self, premask):
        self.n_consumers += 1
        if premask:
            self.n_premask += 1

    def all_premask(self):
        return self.bits is not None and self.n_consumers > 0 and self.n_consumers == self.n_premask


def _masked_in(ctx_info):
    return ctx_info is not None and ctx_info.all_premask()


_NO_PN_EPILOGUE = bool(int(os.environ.get('SARAGAN_NO_PN_EPILOGUE', '0')))   # diagnostic: pixel-norm backward as its own pass


def pn_bwd_epilogue_available(prod_shape, kernel, fmaps, dtype):
    """Whether the data gradient of conv3d(y, [*kernel, c, fmaps]) can apply the backward of the pixel-norm stage that
    produced y in its epilogue (sg_conv_epilogue.pn_bwd_y: sliding-halo kernel, bf16, 3x3x3, 32 channels on both sides,
    32-wide rows).  Decided from the shapes when the graph is built; a launch the library declines after all runs the two
    passes inside the consumer (_dgrad_into)."""
    if _NO_PN_EPILOGUE or dtype != torch.bfloat16 or len(prod_shape) != 5 or tuple(kernel) != (3, 3, 3):
        return False
    n, c, d, h, w = prod_shape
    # (round 5: conv_fwd3w takes this epilogue at any batch -- few columns are cut into D segments; the sliding-halo kernel
    # behind it needed >= 256 column pairs)
    return c == 32 and fmaps == 32 and w % 32 == 0 and d >= 2 and h >= 8


def _dgrad_into(info, g, w, coef, flip):
    """Data gradient of a convolution for an input that is the output of a stage whose backward this consumer has
    registered to apply (ActInfo.all_premask): the LeakyReLU mask in the epilogue, and for a pixel-norm stage its whole
    backward (sg_conv_epilogue.pn_bwd_y) -- the producer then skips its own pass."""
    if info.pn is None:
        if info.pw is not None and not torch.is_grad_enabled() and not _NO_PW_EPILOGUE and g.dtype == torch.bfloat16:
            res = _pw_fused_backward(info, g, w, coef, flip)
            if res is not None:
                return res
        return _Conv.apply(g, w, coef, flip, False, None, info.bits, info.slope)
    if torch.is_grad_enabled():
        raise NotImplementedError('second-order gradient through pixel_norm is not part of the pgan step')
    res = raw_conv(g, w, coef, flip, mask_bits=info.bits, mask_slope=info.slope, pn_bwd=info.pn)
    if res is not None:
        return res[0]
    gy = raw_conv(g, w, coef, flip)[0]          # the library declined (shape): the two passes, here
    return _PnActBwd.apply(gy, info.pn[0], info.pn[1], info.bits, info.slope, False)[0]


_NO_PW_EPILOGUE = bool(int(os.environ.get('SARAGAN_NO_PW_EPILOGUE', '0')))   # diagnostic: from_rgb's backward as its own pass over the gradient


def _pw_fused_backward(info, g, w, coef, flip):
    """conv_1's data gradient with from_rgb's whole backward in its epilogue (sg_conv_epilogue.pw_*): the gradient of from_rgb's
    output -- 32 channels at full resolution, the largest tensor of the discriminator's backward, read once more by
    sg_conv3d_pw_bwd only to be reduced to 32 + 32 numbers and a one-channel image -- is never written.  from_rgb's own
    backward (which autograd runs next) finds (gx, gw, gb) in info.pw_result; what it is handed as `gy` is a stride-0 placeholder.
    None: the library has no such epilogue for this layer."""
    pw = info.pw
    wr = pw['w']
    want_w = wr.data_ptr() not in _SKIP['ptrs']
    want_b = pw['has_b'] and pw['b_ptr'] not in _SKIP['ptrs']
    want_dx = bool(pw['x_req'])
    if not (want_w or want_b or want_dx):
        return None
    x_img = ndhwc(pw['x'])
    if x_img.dtype != g.dtype or x_img.shape[1] != 1:
        return None
    cout = wr.shape[-1]
    dw = _f32_out(wr.data_ptr(), tuple(wr.shape), g.device) if want_w else None
    db = _f32_out(pw['b_ptr'], (cout,), g.device) if want_b else None
    res = raw_conv(g, w, coef, flip, mask_bits=info.bits, mask_slope=info.slope,
                   pw_bwd=dict(x=x_img, wmat=_rgb_matrix(wr, pw['coef'], g.dtype, small_is_cin=True), want_dx=want_dx, dw=dw, db=db,
                               coef=pw['coef']))
    if res is None:
        _unclaim(wr.data_ptr(), dw)
        _unclaim(pw['b_ptr'], db)
        return None
    info.pw_result = (res if want_dx else None, dw, db)
    n, c, d, h, wd = _dims(ndhwc(g))
    return torch.zeros((), device=g.device, dtype=g.dtype).expand(n, w.shape[3] if flip else w.shape[4], d, h, wd)


class BackInfo:
    """The same bookkeeping one order up.  `t = M * conv(x)` (a data-gradient conv with a LeakyReLU mask in its epilogue,
    created while the gradient penalty's first backward builds its graph) has the backward gy -> M * gy, a full pass over
    the tensor.  Every Function that consumes `t` registers here (forward time); consumers that are able to return
    M * (their gradient) from their own kernel -- a conv's mask epilogue -- register as `premask`.  If all of them did,
    they do, and the producer skips its pass (decided at backward time, when all consumers are known).  Functions of this
    module are the only consumers of such tensors; all of them call _note_consumer on their tensor inputs."""

    def __init__(self, bits, slope):
        self.bits, self.slope = bits, float(slope)
        self.n_consumers = 0
        self.n_premask = 0
        self._decision = None

    def all_premask(self):
        # decided once, by whoever asks first (a consumer's backward runs before the producer's): the backward itself
        # hands the tensor to further Functions (the weight gradient of the double backward), which must not change
        # what the consumers that already ran were told
        if self._decision is None:
            self._decision = self.n_consumers > 0 and self.n_consumers == self.n_premask
        return self._decision


_NO_BACK_PREMASK = bool(int(os.environ.get('SARAGAN_NO_BACK_PREMASK', '0')))   # diagnostic: always the separate mask pass


def _note_consumer(t, premask=False):
    info = getattr(t, '_sg_back', None) if t is not None else None
    if info is not None and info._decision is None:
        info.n_consumers += 1
        if premask and not _NO_BACK_PREMASK:
            info.n_premask += 1
    return info


def _note_all(*args):
    """A Function without a mask epilogue consuming the output of a masked conv: counted, never pre-masking."""
    for a in args:
        if isinstance(a, torch.Tensor):
            _note_consumer(a)


class _Conv(torch.autograd.Function):
    """Plain conv3d / dense (networks/ops.py:139-150) or, with flip, its data gradient; `mask_bits` (sign words
    of a tensor shaped like y) fuses a LeakyReLU backward into the epilogue (y *= where(bit, mask_slope, 1))."""

    @staticmethod
    def forward(ctx, x, w, coef, flip, ups, in_info=None, mask_bits=None, mask_slope=0.0):
        ctx.save_for_backward(x, w, mask_bits)
        ctx.coef, ctx.flip, ctx.ups, ctx.in_info, ctx.mask_slope = coef, flip, ups, in_info, mask_slope
        # my backward returns conv'(gy) for x: with a mask epilogue if x is a masked conv's output (see BackInfo)
        ctx.x_back = _note_consumer(x, premask=not ups)
        y, _, _ = raw_conv(x, w, coef, flip, ups, mask_bits=mask_bits, mask_slope=mask_slope)
        ctx.out_back = None
        if mask_bits is not None:
            ctx.out_back = y._sg_back = BackInfo(mask_bits, mask_slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, mask_bits = ctx.saved_tensors
        if mask_bits is not None and not (ctx.out_back is not None and ctx.out_back.all_premask()):
            # y = M * conv(x): pull the (linear) mask back onto the incoming gradient -- unless every consumer of y
            # already did so in the kernel that produced its share of gy
            gy, _ = _BiasActBwd.apply(gy, mask_bits, ctx.mask_slope, False)
        gx = gw = None
        k = tuple(w.shape[:3]) if w.dim() == 5 else (1, 1, 1)
        if ctx.needs_input_grad[0]:
            if ctx.ups:
                gx = _upconv_dgrad(gy, w, ctx.coef, not ctx.flip)
            elif ctx.x_back is not None and ctx.x_back.all_premask():
                gx = _Conv.apply(gy, w, ctx.coef, not ctx.flip, False, None, ctx.x_back.bits, ctx.x_back.slope)
            elif _masked_in(ctx.in_info):
                gx = _dgrad_into(ctx.in_info, gy, w, ctx.coef, not ctx.flip)
            else:
                gx = _Conv.apply(gy, w, ctx.coef, not ctx.flip, False)
        if _wants(ctx, 1, w.data_ptr()):
            if ctx.flip:
                if ctx.ups:
                    raise NotImplementedError
                gw, _ = _Wgrad.apply(gy, x, k, ctx.coef, False, False, w.data_ptr())
            else:
                gw, _ = _Wgrad.apply(x, gy, k, ctx.coef, ctx.ups, False, w.data_ptr())
            gw = gw.reshape(w.shape) if gw is not None else None
        return gx, gw, None, None, None, None, None, None


class _Wgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dy, k, coef, ups, want_db, w_ptr=0, b_ptr=0):
        _note_all(x, dy, k, coef, ups, want_db)
        ctx.save_for_backward(x, dy)
        ctx.k, ctx.coef, ctx.ups = k, coef, ups
        dw, db = raw_wgrad(x, dy, k, coef, ups, want_db, w_ptr, b_ptr)
        if db is None:
            db = _zero_scalar(x)
        ctx.mark_non_differentiable(db)
        return dw, db

    @staticmethod
    def backward(ctx, gw, _gdb):
        x, dy = ctx.saved_tensors
        if ctx.ups:
            raise NotImplementedError('third-order gradient through a fused-upsample conv')
        gx = gdy = None
        if ctx.needs_input_grad[0]:
            gx = _Conv.apply(dy, gw, ctx.coef, True, False)
        if ctx.needs_input_grad[1]:
            gdy = _Conv.apply(x, gw, ctx.coef, False, False)
        return gx, gdy, None, None, None, None, None, None


class _ConvBiasAct(torch.autograd.Function):
    """conv3d -> apply_bias -> act [-> pixel_norm] in one kernel (networks/ops.py:130-150,185-192,308-310).
    out_info / in_info: ActInfo of this layer's output / of the tensor it consumes (see ActInfo)."""

    @staticmethod
    def forward(ctx, x, w, b, coef, ups, act, slope, pixel_norm, eps, out_info=None, in_info=None):
        _note_all(x, w, b, coef, ups, act, slope, pixel_norm, eps, out_info, in_info)
        y, scale, signs = raw_conv(x, w, coef, False, ups, bias=b, act=act, slope=slope, pixel_norm=pixel_norm,
                                   eps=eps, want_scale=True, want_signs=act and not _NO_SIGN_WORDS)
        if out_info is not None:
            out_info.bits = signs
            # (an alias without autograd history: `y` itself will point at this node, which holds out_info -- a reference
            # cycle that kept the stage's output alive until the cyclic collector ran, 1.7 GiB per step at batch 32)
            out_info.pn = (y.detach(), scale) if pixel_norm else None
            if (act and not pixel_norm and not ups and signs is not None and w.dim() == 5 and tuple(w.shape[:4]) == (1, 1, 1, 1) and
                    x.dim() == 5 and not _NO_RGB_FUSION):      # from_rgb on one image channel (pgan/discriminator.py:9-12)
                out_info.pw = dict(x=x.detach(), w=w, coef=coef, b_ptr=b.data_ptr() if b is not None else 0, has_b=b is not None,
                                   x_req=x.requires_grad)
        ctx.save_for_backward(x, w, y if (pixel_norm or (act and signs is None)) else None, scale, signs)
        ctx.cfg = (coef, ups, act, slope, pixel_norm)
        ctx.has_b = b is not None
        ctx.b_ptr = b.data_ptr() if b is not None else 0
        ctx.out_info, ctx.in_info = out_info, in_info
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, scale, signs = ctx.saved_tensors
        coef, ups, act, slope, pixel_norm = ctx.cfg
        if ctx.out_info is not None and ctx.out_info.pw_result is not None:
            # from_rgb: the consumer's data-gradient kernel ran this whole backward in its epilogue (_pw_fused_backward); `gy` is a placeholder
            gx, gw, gb = ctx.out_info.pw_result
            ctx.out_info.pw_result = None
            return (gx if ctx.needs_input_grad[0] else None), gw, gb, None, None, None, None, None, None, None, None
        g = gy
        want_db = ctx.has_b and _wants(ctx, 2, ctx.b_ptr)
        gb = None
        fused_pn_act = pixel_norm and act and signs is not None
        if fused_pn_act and _masked_in(ctx.out_info):
            pass        # every consumer pushed its share of gy through this stage's backward already (_dgrad_into)
        elif fused_pn_act:
            g, gb = _PnActBwd.apply(g, y.detach(), scale, signs, slope, want_db, ctx.b_ptr)
            if not want_db:
                gb = None
        elif pixel_norm:
            g = _PixelNormBwd.apply(g, y, scale)
        premasked = act and not pixel_norm and _masked_in(ctx.out_info)   # every consumer already applied my mask
        if act and not premasked and not fused_pn_act:
            g, gb = _BiasActBwd.apply(g, signs if signs is not None else y.detach(), slope, want_db, ctx.b_ptr)
        gx = gw = None
        db_from_wgrad = want_db and gb is None
        if (ctx.needs_input_grad[0] and _wants(ctx, 1, w.data_ptr()) and not torch.is_grad_enabled() and not ups and
                not _masked_in(ctx.in_info) and w.dim() == 5 and x.dim() == 5 and tuple(w.shape[:3]) == (1, 1, 1) and w.shape[3] <= 4 and
                not _NO_RGB_FUSION):
            res = _pw_backward(x, g, w, coef, db_from_wgrad, ctx.b_ptr)      # from_rgb: one pass over its output gradient
            if res is not None:
                gx, gw, gb2 = res
                return gx, gw, ((gb2 if db_from_wgrad else gb) if want_db else None), None, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            if ups:
                gx = _upconv_dgrad(g, w, coef, True)
            elif _masked_in(ctx.in_info):
                gx = _dgrad_into(ctx.in_info, g, w, coef, True)
            else:
                gx = _Conv.apply(g, w, coef, True, False)
        if _wants(ctx, 1, w.data_ptr()):
            k = tuple(w.shape[:3]) if w.dim() == 5 else (1, 1, 1)
            gw, gb2 = _Wgrad.apply(x, g, k, coef, ups, db_from_wgrad, w.data_ptr(), ctx.b_ptr)
            gw = gw.reshape(w.shape) if gw is not None else None
            if db_from_wgrad:
                gb = gb2
        elif db_from_wgrad:
            _, gb = raw_bias_act_bwd(g, None, 0.0, want_dx=False, want_db=True, b_ptr=ctx.b_ptr)
        return gx, gw, (gb if want_db else None), None, None, None, None, None, None, None, None


_NO_RGB_FWD_EPILOGUE = bool(int(os.environ.get('SARAGAN_NO_RGB_FWD_EPILOGUE', '0')))   # diagnostic: to_rgb's forward as its own pass over y
_NO_RGB_FUSION = bool(int(os.environ.get('SARAGAN_NO_RGB_FUSION', '0')))   # diagnostic: to_rgb's data gradient as a tensor
_NO_RGB_WG_FUSION = bool(int(os.environ.get('SARAGAN_NO_RGB_WG_FUSION', '0')))   # diagnostic: to_rgb's filter gradient as its own pass over y


def _rgb_matrix(w_rgb, coef, dtype, small_is_cin=False):
    """[cs][c] f32 of a pointwise convolution between cs <= 4 and c channels: the values its forward multiplied with
    (coef * w rounded to the compute dtype, as the packed weight image holds them), for sg_pixel_norm_act_bwd_pw (to_rgb:
    cs = cout) and sg_conv3d_pw_bwd (from_rgb: cs = cin).  Cached like the packed images."""
    key = ('rgbmat', w_rgb.data_ptr(), w_rgb._version, float(coef), dtype, small_is_cin)
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit[0]
    m = (w_rgb.detach().reshape(w_rgb.shape[-2], w_rgb.shape[-1]).float() * float(coef)).to(dtype).float()
    m = (m if small_is_cin else m.t()).contiguous()
    _PACK_CACHE[key] = (m, w_rgb)
    return m


def _pw_backward(x, dy, w, coef, want_db, b_ptr=0):
    """(gx, gw, gb) of a pointwise convolution from <= 4 input channels (from_rgb) in one pass over dy
    (sg_conv3d_pw_bwd), or None where the library has no such pass."""
    lib = _lib.load()
    x, dy = ndhwc(x), ndhwc(dy)
    n, cin, d, h, wd = _dims(x)
    cout = dy.shape[1]
    if x.dtype != dy.dtype or cin > 4 or cin > cout:
        return None
    shp = _shape(n, d, h, wd, cin, cout, (1, 1, 1), False)
    dt = _dt(x)
    ws_bytes = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    dw = _f32_out(w.data_ptr(), (1, 1, 1, cin, cout), x.device)
    db = _f32_out(b_ptr, (cout,), x.device) if want_db else None
    gx = torch.empty_like(x)
    rc = lib.sg_conv3d_pw_bwd(_ptr(x), _ptr(dy), _ptr(_rgb_matrix(w, coef, x.dtype, True)), _ptr(dw), _ptr(db), _ptr(gx),
                              float(coef), _ptr(ws), ws_bytes, C.byref(shp), dt, _stream())
    if rc == _lib.SG_EUNSUPPORTED:
        _unclaim(w.data_ptr(), dw)
        _unclaim(b_ptr, db)
        return None
    check(rc, 'sg_conv3d_pw_bwd')
    return gx, dw.reshape(w.shape), db


class _ConvPnActToRgb(torch.autograd.Function):
    """One generator stage's tail, y = pixel_norm(leaky_relu(conv3d(x) + b)) followed by img = to_rgb(y)
    (pgan/generator.py:33-45,96-97), as one node with both outputs.  Forward is the two launches it would be anyway.
    Backward, when y has no other consumer (the last block: only to_rgb reads it): to_rgb's data gradient -- c channels
    at full resolution, the largest tensor of G's backward -- is never written; sg_pixel_norm_act_bwd_pw forms it in
    registers from the 1-channel image gradient inside the pixel-norm / LeakyReLU backward pass.  Once-differentiable,
    like _PnActBwd."""

    @staticmethod
    def forward(ctx, x, w, b, coef, ups, slope, eps, in_info, w_rgb, b_rgb, coef_rgb):
        _note_all(x, w, b, w_rgb, b_rgb)
        res = None
        if (w_rgb.shape[-1] == 1 and not ups and x.dtype == torch.bfloat16 and not _NO_RGB_FWD_EPILOGUE and
                (b_rgb is None or b_rgb.dtype == torch.float32)):
            # to_rgb inside the stage's own epilogue (sg_conv_epilogue.rgb_*): the stage's output is not read again for the image
            res = raw_conv(x, w, coef, False, ups, bias=b, act=True, slope=slope, pixel_norm=True, eps=eps, want_scale=True,
                           want_signs=True, rgb=(_rgb_matrix(w_rgb, coef_rgb, x.dtype), b_rgb))
        if res is not None:
            y, scale, signs, img = res
        else:
            y, scale, signs = raw_conv(x, w, coef, False, ups, bias=b, act=True, slope=slope, pixel_norm=True, eps=eps,
                                       want_scale=True, want_signs=True)
            img, _, _ = raw_conv(y, w_rgb, coef_rgb, False, False, bias=b_rgb)
        ctx.save_for_backward(x, w, y, scale, signs, w_rgb)
        ctx.cfg = (coef, ups, slope, coef_rgb)
        ctx.in_info = in_info
        ctx.ptrs = (w.data_ptr(), b.data_ptr() if b is not None else 0, w_rgb.data_ptr(),
                    b_rgb.data_ptr() if b_rgb is not None else 0)
        ctx.has_b, ctx.has_b_rgb = b is not None, b_rgb is not None
        ctx.set_materialize_grads(False)
        return img, y

    @staticmethod
    def backward(ctx, g_img, g_y):
        if torch.is_grad_enabled():
            raise NotImplementedError('second-order gradient through pixel_norm is not part of the pgan step')
        x, w, y, scale, signs, w_rgb = ctx.saved_tensors
        coef, ups, slope, coef_rgb = ctx.cfg
        lib = _lib.load()
        n, c, d, h, wd = _dims(y)
        nvox = n * d * h * wd
        want_db = ctx.has_b and _wants(ctx, 2, ctx.ptrs[1])
        gw_rgb = gb_rgb = None
        if g_img is None and g_y is None:
            return (None,) * 11
        g = gb = None
        cs = w_rgb.shape[-1]
        fuse = g_img is not None and g_y is None and not _NO_RGB_FUSION and cs <= 4
        want_w_rgb = g_img is not None and _wants(ctx, 8, ctx.ptrs[2])
        want_db_rgb = g_img is not None and ctx.has_b_rgb and _wants(ctx, 9, ctx.ptrs[3])
        if g_img is not None:
            g_img = ndhwc(g_img)
        done_rgb = False
        if fuse and cs == 1 and (want_w_rgb or want_db_rgb) and not _NO_RGB_WG_FUSION:
            # to_rgb's data gradient AND its own filter / bias gradient inside the pixel-norm / LeakyReLU backward pass: all three
            # need only y and the image gradient, which that pass reads anyway (a separate filter-gradient pass re-read y: 1.07 GB)
            g = torch.empty_like(y)
            gb = _f32_out(ctx.ptrs[1], (c,), y.device) if want_db else None
            gw_rgb = _f32_out(ctx.ptrs[2], tuple(w_rgb.shape), y.device) if want_w_rgb else None
            gb_rgb = _f32_out(ctx.ptrs[3], (cs,), y.device) if want_db_rgb else None
            ws_bytes = lib.sg_pixel_norm_act_bwd_pw_wg_workspace(c, cs)
            ws = torch.empty(ws_bytes, device=y.device, dtype=torch.uint8)
            rc = lib.sg_pixel_norm_act_bwd_pw_wg(_ptr(g_img), cs, _ptr(_rgb_matrix(w_rgb, coef_rgb, y.dtype)), _ptr(y), _ptr(scale), _ptr(signs),
                                                 float(slope), _ptr(g), _ptr(gb), _ptr(gw_rgb), _ptr(gb_rgb), float(coef_rgb), _ptr(ws), ws_bytes,
                                                 nvox, c, _dt(y), _stream())
            if rc == _lib.SG_EUNSUPPORTED:
                _unclaim(ctx.ptrs[1], gb)
                _unclaim(ctx.ptrs[2], gw_rgb)
                _unclaim(ctx.ptrs[3], gb_rgb)
                g = gb = gw_rgb = gb_rgb = None
            else:
                check(rc, 'sg_pixel_norm_act_bwd_pw_wg')
                done_rgb = True
        if g_img is not None and not done_rgb:
            if want_w_rgb:
                gw_rgb, gb_rgb = raw_wgrad(y, g_img, (1, 1, 1), coef_rgb, False, want_db_rgb, ctx.ptrs[2], ctx.ptrs[3])
                gw_rgb = gw_rgb.reshape(w_rgb.shape) if gw_rgb is not None else None
            elif want_db_rgb:
                _, gb_rgb = raw_bias_act_bwd(g_img, None, 0.0, want_dx=False, want_db=True, b_ptr=ctx.ptrs[3])
        if fuse and g is None:
            g = torch.empty_like(y)
            gb = _f32_out(ctx.ptrs[1], (c,), y.device) if want_db else None
            ws = torch.empty(lib.sg_bias_act_bwd_workspace(c), device=y.device, dtype=torch.uint8) if want_db else None
            rc = lib.sg_pixel_norm_act_bwd_pw(_ptr(g_img), cs, _ptr(_rgb_matrix(w_rgb, coef_rgb, y.dtype)), _ptr(y), _ptr(scale),
                                              _ptr(signs), float(slope), _ptr(g), _ptr(gb), _ptr(ws), nvox, c, _dt(y), _stream())
            if rc == _lib.SG_EUNSUPPORTED:
                g = None
                _unclaim(ctx.ptrs[1], gb)
            else:
                check(rc, 'sg_pixel_norm_act_bwd_pw')
        if g is None:       # y has other consumers (or the library declined): the gradient for y as a tensor
            gy = raw_conv(g_img, w_rgb, coef_rgb, True, False)[0] if g_img is not None else None
            if g_y is not None:
                gy = ndhwc(g_y) if gy is None else gy + g_y
            g, gb = _PnActBwd.apply(gy, y, scale, signs, slope, want_db, ctx.ptrs[1])
            if not want_db:
                gb = None
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if ups:
                gx = _upconv_dgrad(g, w, coef, True)
            elif _masked_in(ctx.in_info):
                gx = _dgrad_into(ctx.in_info, g, w, coef, True)
            else:
                gx = raw_conv(g, w, coef, True, False)[0]
        if _wants(ctx, 1, ctx.ptrs[0]):
            k = tuple(w.shape[:3]) if w.dim() == 5 else (1, 1, 1)
            gw, _ = raw_wgrad(x, g, k, coef, ups, False, ctx.ptrs[0])
            gw = gw.reshape(w.shape) if gw is not None else None
        return gx, gw, gb, None, None, None, None, None, gw_rgb, gb_rgb, None


_NO_POOL3 = bool(int(os.environ.get('SARAGAN_NO_POOL3', '0')))   # diagnostic: D x W means from the epilogue + the H pairs in a pass of their own
_NO_POOL_FUSION = bool(int(os.environ.get('SARAGAN_NO_POOL_FUSION', '0')))   # diagnostic: conv and downscale3d apart


class _ConvBiasActPool(torch.autograd.Function):
    """downscale3d(leaky_relu(conv3d(x) + b)) (one discriminator block's tail, pgan/discriminator.py:33-44) without the
    full-resolution activation: the convolution's epilogue writes the means of 2 x 1 x 2 blocks (sg_conv_epilogue.pool = 1,
    layers with <= 32 input channels) or 1 x 2 x 2 blocks (pool = 2) and the sign words; sg_downscale_sum pools the pairs
    that are left.  Backward: the pooled gradient goes up through ONE masked nearest-x2 (gain 1/8, this layer's sign
    words) -- written as two 32-channel tensors when the layer has 64 outputs and nothing differentiates the backward
    again (_pooled_backward_planes) -- then the usual data / weight gradients."""

    @staticmethod
    def forward(ctx, x, w, b, coef, slope, in_info=None):
        _note_all(x, w, b, coef, slope, in_info)
        mode = _pool_mode(x, w.shape[:3], x.shape[1], w.shape[-1]) if w.dim() == 5 else 0
        res = None
        if mode == 3:      # the whole 2 x 2 x 2 mean from the epilogue (sg_conv_epilogue.pool = 3): one rounding
            res = raw_conv(x, w, coef, False, False, bias=b, act=True, slope=slope, want_signs=True, pool=3)
            mode = 1
        if res is not None:
            y, _, signs = res
        else:
            res = raw_conv(x, w, coef, False, False, bias=b, act=True, slope=slope, want_signs=True, pool=mode) if mode else None
            if res is None:
                raise _lib.SgError('pool fusion not available for this layer')
            ydw, _, signs = res
            y = _Down.apply(ydw, 0.5, None, _POOL_REST[mode])
        ctx.save_for_backward(x, w, signs)
        ctx.cfg = (coef, slope)
        ctx.has_b = b is not None
        ctx.b_ptr = b.data_ptr() if b is not None else 0
        ctx.in_info = in_info
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, signs = ctx.saved_tensors
        coef, slope = ctx.cfg
        want_db = ctx.has_b and _wants(ctx, 2, ctx.b_ptr)
        if not torch.is_grad_enabled():
            res = None
            if not _NO_GATHER_BWD:
                res = _pooled_backward_gather(gy, x, w, signs, coef, slope, ctx.in_info, ctx.needs_input_grad[0],
                                              _wants(ctx, 1, w.data_ptr()), want_db, ctx.b_ptr)
            if res is None and not _NO_PLANES:
                res = _pooled_backward_planes(gy, x, w, signs, coef, slope, ctx.in_info, ctx.needs_input_grad[0],
                                              _wants(ctx, 1, w.data_ptr()), want_db)
            if res is not None:
                return res[0], res[1], (res[2] if want_db else None), None, None, None
        if (torch.is_grad_enabled() and not _NO_GATHER_BWD and ctx.needs_input_grad[0] and not _wants(ctx, 1, w.data_ptr()) and
                not want_db and gy.dtype == torch.bfloat16 and w.dim() == 5 and tuple(w.shape) == (3, 3, 3, 32, 64) and signs is not None and
                not (_masked_in(ctx.in_info) and ctx.in_info.pn is not None)):
            # the gradient penalty's first backward: only the data gradient is wanted, and it is differentiated again
            masked = _masked_in(ctx.in_info)
            try:
                gx = _PooledDgradGather.apply(gy, w, signs, slope, coef, ctx.in_info.bits if masked else None,
                                              ctx.in_info.slope if masked else 0.0)
                return gx, None, None, None, None, None
            except _GatherDeclined:
                pass
        g = _Up.apply(gy, 0.125, signs, slope, (2, 2, 2))      # d(downscale3d) * LeakyReLU mask, full resolution
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if _masked_in(ctx.in_info):
                gx = _dgrad_into(ctx.in_info, g, w, coef, True)
            else:
                gx = _Conv.apply(g, w, coef, True, False)
        if _wants(ctx, 1, w.data_ptr()):
            gw, gb = _Wgrad.apply(x, g, tuple(w.shape[:3]), coef, False, want_db, w.data_ptr(), ctx.b_ptr)
            gw = gw.reshape(w.shape) if gw is not None else None
        elif want_db:
            _, gb = raw_bias_act_bwd(g, None, 0.0, want_dx=False, want_db=True, b_ptr=ctx.b_ptr)
        return gx, gw, (gb if want_db else None), None, None, None


_NO_PLANES = bool(int(os.environ.get('SARAGAN_NO_PLANES', '0')))   # diagnostic: the 64-channel gradient as one tensor
_NO_GATHER_BWD = bool(int(os.environ.get('SARAGAN_NO_GATHER_BWD', '0')))   # diagnostic: materialise the up-scaled gradient


class _GatherDeclined(Exception):
    """The library has no fused-gather tile for this launch (the caller takes the materialised path)."""


def _gather_dgrad_launch(gy, w, signs, slope, coef, mask_bits, mask_slope):
    """gx = [mask] conv'(M * upscale3d(gy) / 8) through sg_conv_epilogue.in_mask_bits (the two-pass 64 -> 32 path)."""
    lib = _lib.load()
    gy = ndhwc(gy)
    n, cout, dc, hc, wc = _dims(gy)
    d, h, wd = 2 * dc, 2 * hc, 2 * wc
    dt, st = _dt(gy), _stream()
    shp = _shape(n, d, h, wd, 64, 32, (3, 3, 3), True)
    ws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
    if not ws_bytes:
        raise _GatherDeclined()
    _check_signs(signs, n * d * h * wd, 64)
    wp = _packed(w, coef, True, shp, dt, lib, st)
    gx = _empty_like_shape(gy, 32, (d, h, wd))
    ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, _ptr(mask_bits), float(mask_slope) if mask_bits is not None else 0.0, None)
    if mask_bits is not None:
        _check_signs(mask_bits, n * d * h * wd, 32)
    ws = torch.empty(ws_bytes, device=gy.device, dtype=torch.uint8)
    ep.workspace, ep.workspace_bytes = ws.data_ptr(), ws_bytes
    ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = signs.data_ptr(), float(slope), 0.125
    rc = lib.sg_conv3d_fwd(_ptr(gy), _ptr(wp), _ptr(gx), C.byref(shp), C.byref(ep), dt, st)
    if rc == _lib.SG_EUNSUPPORTED:
        raise _GatherDeclined()
    check(rc, 'sg_conv3d_fwd (masked gather)')
    return gx


class _PooledDgradGather(torch.autograd.Function):
    """Data gradient of downscale3d(leaky_relu(conv3d(x) + b)) for x from the POOLED gradient gy, differentiable once more
    (the gradient penalty's first backward, networks/loss.py:136-140): gx = [M_in *] conv'(M * upscale3d(gy) / 8), the masked
    up-scale formed in the convolution's gather instead of being written (and kept for the double backward) as a tensor.
    Backward: for gy, block_sum(M * conv(ggx)) / 8 -- the layer's own forward on the incoming gradient; for w, the weight
    gradient of (ggx, M * upscale3d(gy) / 8), gathered the same way (sg_conv3d_wgrad_bias_up_masked)."""

    @staticmethod
    def forward(ctx, gy, w, signs, slope, coef, mask_bits, mask_slope):
        _note_all(gy)
        gx = _gather_dgrad_launch(gy, w, signs, slope, coef, mask_bits, mask_slope)
        ctx.save_for_backward(gy, w, signs, mask_bits)
        ctx.cfg = (float(slope), float(coef), float(mask_slope))
        ctx.out_back = None
        if mask_bits is not None:
            ctx.out_back = gx._sg_back = BackInfo(mask_bits, mask_slope)
        return gx

    @staticmethod
    def backward(ctx, ggx):
        gy, w, signs, mask_bits = ctx.saved_tensors
        slope, coef, mask_slope = ctx.cfg
        if mask_bits is not None and not (ctx.out_back is not None and ctx.out_back.all_premask()):
            ggx, _ = _BiasActBwd.apply(ggx, mask_bits, mask_slope, False)      # pull the (linear) output mask back
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            # block_sum(M * conv(ggx)) / 8: the mask and the 2 x 1 x 2 block means in the convolution's epilogue
            # (sg_conv_epilogue.pool with mask_bits), the H pairs by sg_downscale_sum -- the 64-channel tensor is never written
            res = res3 = None
            mode = _pool_mode(ggx, (3, 3, 3), 32, 64) if not torch.is_grad_enabled() else 0
            if mode in (1, 3):
                if mode == 3:
                    res3 = raw_conv(ggx, w, coef, False, False, mask_bits=signs, mask_slope=slope, pool=3)
                if res3 is None:
                    res = raw_conv(ggx, w, coef, False, False, mask_bits=signs, mask_slope=slope, pool=1)
            if res3 is not None:
                g_gy = res3[0]
            elif res is not None:
                g_gy = _Down.apply(res[0], 0.5, None, _POOL_REST[1])
            else:
                g_gy = _Down.apply(_Conv.apply(ggx, w, coef, False, False), 0.125, None, (2, 2, 2), signs, slope)
        if _wants(ctx, 1, w.data_ptr()):
            lib = _lib.load()
            ggx_, gy_ = ndhwc(ggx), ndhwc(gy)
            n, _, d, h, wd = _dims(ggx_)
            dt = _dt(ggx_)
            shp = _shape(n, d, h, wd, 32, 64, (3, 3, 3), False)
            acc = _grad_acc(w.data_ptr(), (3, 3, 3, 32, 64))
            rc = _lib.SG_EUNSUPPORTED
            if acc is not None:      # the first-order contribution is in the slot already: this one is added there
                rc = _wgrad_launch(lib, ggx_, gy_, signs, slope, 0.125, acc, None, coef, True, shp, dt)
            if rc != _lib.SG_EUNSUPPORTED:
                check(rc, 'sg_conv3d_wgrad_bias_ex (gathered dy, accumulate)')
                _note_accumulated(w.data_ptr())
                return g_gy, None, None, None, None, None, None
            dw = _f32_out(w.data_ptr(), (3, 3, 3, 32, 64), ggx.device)
            rc = _wgrad_launch(lib, ggx_, gy_, signs, slope, 0.125, dw, None, coef, False, shp, dt)
            if rc == _lib.SG_EUNSUPPORTED:      # the materialised pair, as the plain path computes it
                _unclaim(w.data_ptr(), dw)
                g_full = _Up.apply(gy_, 0.125, signs, slope, (2, 2, 2))
                dw, _ = raw_wgrad(ggx_, g_full, (3, 3, 3), coef, False, False, w.data_ptr())
            else:
                check(rc, 'sg_conv3d_wgrad_bias_ex (gathered dy)')
            g_w = dw.reshape(w.shape) if dw is not None else None
        return g_gy, g_w, None, None, None, None, None


def _pooled_backward_gather(gy, x, w, signs, coef, slope, in_info, want_gx, want_gw, want_db, b_ptr=0):
    """Backward of _ConvBiasActPool for the 32 -> 64 layer when nothing differentiates it again, WITHOUT the up-scaled
    gradient: M * upscale3d(gy) / 8 (64 channels at full resolution, 4.3 GB at batch 64 -- written once and read twice
    by _pooled_backward_planes) is formed from the pooled gradient and the layer's sign words while the consumers stage their
    tiles: the two-pass 64 -> 32 data gradient (sg_conv_epilogue.in_mask_bits with upsample_in) and the weight / bias
    gradient (sg_conv3d_wgrad_bias_up_masked).  Bit-identical to the materialised path.  Returns (gx, gw, gb) or None
    when the layer is not of that shape / the library declines."""
    if gy.dtype != torch.bfloat16 or w.dim() != 5 or tuple(w.shape) != (3, 3, 3, 32, 64) or signs is None:
        return None
    lib = _lib.load()
    gy, x = ndhwc(gy), ndhwc(x)
    n, cout, dc, hc, wc = _dims(gy)
    d, h, wd = 2 * dc, 2 * hc, 2 * wc
    dt, st = _dt(gy), _stream()
    k = (3, 3, 3)
    _check_signs(signs, n * d * h * wd, 64)
    gx = gw = gb = None
    if want_gx:
        shp = _shape(n, d, h, wd, 64, 32, k, True)
        ws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
        if not ws_bytes:
            return None
        wp = _packed(w, coef, True, shp, dt, lib, st)
        gx = _empty_like_shape(gy, 32, (d, h, wd))
        masked = _masked_in(in_info)
        ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, _ptr(in_info.bits) if masked else None,
                          float(in_info.slope) if masked else 0.0, None)
        if masked:
            _check_signs(in_info.bits, n * d * h * wd, 32)
        ws = torch.empty(ws_bytes, device=gy.device, dtype=torch.uint8)
        ep.workspace, ep.workspace_bytes = ws.data_ptr(), ws_bytes
        ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = signs.data_ptr(), float(slope), 0.125
        rc = lib.sg_conv3d_fwd(_ptr(gy), _ptr(wp), _ptr(gx), C.byref(shp), C.byref(ep), dt, st)
        if rc == _lib.SG_EUNSUPPORTED:
            return None
        check(rc, 'sg_conv3d_fwd (masked gather)')
    if want_gw or want_db:
        shp = _shape(n, d, h, wd, 32, 64, k, False)
        acc = _grad_acc(w.data_ptr(), (3, 3, 3, 32, 64)) if want_gw else None
        dw = acc if acc is not None else _f32_out(w.data_ptr() if want_gw else 0, (3, 3, 3, 32, 64), gy.device)
        gb = _f32_out(b_ptr, (64,), gy.device) if want_db else None
        rc = _wgrad_launch(lib, x, gy, signs, slope, 0.125, dw, gb, coef, acc is not None, shp, dt)
        if rc == _lib.SG_EUNSUPPORTED:
            if acc is None:
                _unclaim(w.data_ptr(), dw)
            _unclaim(b_ptr, gb)
            return None
        check(rc, 'sg_conv3d_wgrad_bias_ex (gathered dy)')
        if acc is not None:
            _note_accumulated(w.data_ptr())
        gw = dw.reshape(w.shape) if (want_gw and acc is None) else None
    return gx, gw, gb


def _pooled_backward_planes(gy, x, w, signs, coef, slope, in_info, want_gx, want_gw, want_db):
    """Backward of _ConvBiasActPool for a 64-channel layer when nothing differentiates it again: the masked, up-scaled
    gradient (64 channels at full resolution, the largest tensor of the backward pass) is written as two 32-channel
    tensors (sg_upscale_nn_planes).  The data gradient -- a 64 -> 32 convolution, two passes over 32 input channels each
    with f32 partial sums -- then reads whole 64-byte rows (sg_conv_epilogue.x_plane_channels) where the interleaved
    layout made it fetch every 128-byte line twice, and the weight gradient is one 32-channel launch per tensor.
    Returns (gx, gw, gb) or None when the layer is not of that shape / the library declines."""
    if gy.dtype != torch.bfloat16 or w.dim() != 5 or tuple(w.shape[:3]) != (3, 3, 3) or w.shape[4] != 64 or w.shape[3] != 32:
        return None
    lib = _lib.load()
    gy = ndhwc(gy)
    n, cout, dc, hc, wc = _dims(gy)
    d, h, wd = 2 * dc, 2 * hc, 2 * wc
    dt, st = _dt(gy), _stream()
    k = (3, 3, 3)
    shp = _shape(n, d, h, wd, 64, 32, k, False)
    ws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
    if not ws_bytes or signs is None:
        return None
    _check_signs(signs, n * d * h * wd, 64)
    planes = torch.empty((2, n, d, h, wd, 32), device=gy.device, dtype=gy.dtype)
    rc = lib.sg_upscale_nn_planes(_ptr(gy), _ptr(planes), _ptr(signs), float(slope), n, dc, hc, wc, 64, 2, 2, 2, 0.125, 32,
                                  dt, st)
    if rc == _lib.SG_EUNSUPPORTED:
        return None
    check(rc, 'sg_upscale_nn_planes')
    gx = gw = gb = None
    if want_gx:
        wp = _packed(w, coef, True, shp, dt, lib, st)
        gx = _empty_like_shape(gy, 32, (d, h, wd))
        masked = _masked_in(in_info)
        ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, _ptr(in_info.bits) if masked else None,
                          float(in_info.slope) if masked else 0.0, None)
        if masked:
            _check_signs(in_info.bits, n * d * h * wd, 32)
        ws = torch.empty(ws_bytes, device=gy.device, dtype=torch.uint8)
        ep.workspace, ep.workspace_bytes, ep.x_plane_channels = ws.data_ptr(), ws_bytes, 32
        rc = lib.sg_conv3d_fwd(_ptr(planes), _ptr(wp), _ptr(gx), C.byref(shp), C.byref(ep), dt, st)
        if rc == _lib.SG_EUNSUPPORTED:
            return None
        check(rc, 'sg_conv3d_fwd (planes)')
    if want_gw or want_db:
        halves = [planes[i].permute(0, 4, 1, 2, 3) for i in range(2)]       # [n,32,d,h,w], channels last
        if want_gw:
            parts = [raw_wgrad(x, g, k, coef, False, want_db) for g in halves]
            gw = torch.cat([p[0] for p in parts], dim=4).reshape(w.shape)
            if want_db:
                gb = torch.cat([p[1] for p in parts])
        else:
            gb = torch.cat([raw_bias_act_bwd(g, None, 0.0, want_dx=False, want_db=True)[1] for g in halves])
    return gx, gw, gb


def _pool_mode(x, k, cin, cout):
    """Which sg_conv_epilogue.pool mode may fuse the 2x2x2 pooling of this convolution's output (0: none), from the
    shape; the library has the last word (SG_EUNSUPPORTED -> raw_conv returns None)."""
    if _NO_POOL_FUSION or x.dim() != 5 or x.dtype != torch.bfloat16 or tuple(k) != (3, 3, 3):
        return 0
    n, _, d, h, wd = x.shape
    if (d | h | wd) & 1 or wd % 32:
        return 0
    nvox = n * d * h * wd
    if cin <= 32 and cin % 8 == 0 and cout % 32 == 0 and d >= 4 and nvox >= (1 << 20):
        if cin == 32 and h >= 8 and not _NO_POOL3:
            return 3    # wave-private planes (conv_fwd3w): the whole 2 x 2 x 2 mean in the epilogue, rounded once
        return 1        # sliding-halo kernel: D x W pairs in the epilogue
    if cin % 16 == 0 and cout % 64 == 0 and nvox >= (1 << 18):
        return 2        # streamed ping-pong kernel: H x W pairs
    return 0


_POOL_REST = {1: (1, 2, 1), 2: (2, 1, 1)}     # the pairs left to sg_downscale_sum after the epilogue's block means


def _upconv_dgrad_subpixel(g, w, coef):
    """The same gradient in sub-pixel form (sg_upconv3d_subpixel_dgrad): one launch on the low-resolution grid, 64 tap
    products per voxel with the forward's summed weights transposed, one rounding.  None where the library has no tile."""
    lib = _lib.load()
    g = ndhwc(g)
    n, co, d2, h2, w2 = _dims(g)
    ci = w.shape[3]
    if (d2 | h2 | w2) & 1 or w.shape[4] != co:
        return None
    dt, st = _dt(g), _stream()
    low = _shape(n, d2 // 2, h2 // 2, w2 // 2, ci, co, (3, 3, 3), False)
    if not lib.sg_upconv3d_subpixel_dgrad_supported(C.byref(low), dt):
        return None
    key = ('subpix_dgrad', w.data_ptr(), w._version, float(coef), dt)
    hit = _SUBPIX_CACHE.get(key)
    if hit is None:
        w32 = w.detach()
        if w32.dtype != torch.float32 or not w32.is_contiguous():
            w32 = w32.contiguous().float()
        wp = torch.empty(lib.sg_upconv3d_subpixel_dgrad_packed_bytes(C.byref(low), dt), device=w.device, dtype=torch.uint8)
        check(lib.sg_upconv3d_subpixel_dgrad_pack(_ptr(w32), float(coef), _ptr(wp), C.byref(low), dt, st), 'sg_upconv3d_subpixel_dgrad_pack')
        if w.is_leaf or not w.requires_grad:
            _SUBPIX_CACHE[key] = (wp, w)
    else:
        wp = hit[0]
    gx = _empty_like_shape(g, ci, (d2 // 2, h2 // 2, w2 // 2))
    rc = lib.sg_upconv3d_subpixel_dgrad(_ptr(g), _ptr(wp), _ptr(gx), C.byref(low), dt, st)
    if rc == _lib.SG_EUNSUPPORTED:
        return None
    check(rc, 'sg_upconv3d_subpixel_dgrad')
    return gx


def _upconv_dgrad(g, w, coef, flip):
    """Gradient of conv3d(upscale3d(x)) (pgan/generator.py:33-34) for x: the 2x2x2 block SUM of the data gradient.  When
    nothing differentiates this backward again, the sliding-halo kernel pools 2 x 1 x 2 in its epilogue
    (sg_conv_epilogue.pool) and the full-resolution data gradient -- cin x the fine volume -- is never written."""
    if not torch.is_grad_enabled() and w.dim() == 5:
        if flip and not _NO_SUBPIXEL and tuple(w.shape[:3]) == (3, 3, 3) and g.dtype == torch.bfloat16:
            gx = _upconv_dgrad_subpixel(g, w, coef)
            if gx is not None:
                return gx
        cin, cout = (w.shape[4], w.shape[3]) if flip else (w.shape[3], w.shape[4])
        mode = min(_pool_mode(g, w.shape[:3], cin, cout), 2)      # (3: this caller keeps the D x W means + the H pairs)
        if mode:
            res = raw_conv(g, w, coef, flip, False, pool=mode)
            if res is not None:      # means over four voxels; the remaining pairs and the factor back to a sum follow
                return _Down.apply(res[0], 4.0, None, _POOL_REST[mode])
    return _Down.apply(_Conv.apply(g, w, coef, flip, False), 1.0)


def pool_fusion_available(x, w, bias, slope):
    """Whether conv3d + bias + LeakyReLU + downscale3d of this layer runs as _ConvBiasActPool (bf16, 3x3x3, <= 32 input
    channels, 32-wide rows, whole 32-channel output tiles: sg_conv_epilogue.pool).  Decided from the shape; the library
    has the last word (SG_EUNSUPPORTED)."""
    return w.dim() == 5 and _pool_mode(x, w.shape[:3], x.shape[1] if x.dim() == 5 else 0, w.shape[-1]) != 0


class _BiasActBwd(torch.autograd.Function):
    """dx = where(y >= 0, dy, slope*dy) (networks/ops.py:177), db = sum_v dx.  Linear in dy: its own
    backward is the same mask again (networks/ops.py:178).  `y` is the activation or its int32 sign words."""

    @staticmethod
    def forward(ctx, dy, y, slope, want_db, b_ptr=0):
        _note_all(dy, y, slope, want_db)
        dx, db = raw_bias_act_bwd(dy, y, slope, True, want_db, b_ptr)
        ctx.save_for_backward(y)
        ctx.slope = slope
        if db is None:
            db = _zero_scalar(dy)
        ctx.mark_non_differentiable(db)
        return dx, db

    @staticmethod
    def backward(ctx, gdx, _gdb):
        (y,) = ctx.saved_tensors
        g, _ = _BiasActBwd.apply(gdx, y, ctx.slope, False)
        return g, None, None, None, None


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, act, slope):
        _note_all(x, b, act, slope)
        lib = _lib.load()
        _req_cuda(x, b)
        x = ndhwc(x)
        n, c, d, h, w = _dims(x)
        y = torch.empty_like(x)
        b32 = b.detach().contiguous().float() if b is not None else None
        check(lib.sg_bias_act_fwd(_ptr(x), _ptr(b32), _ptr(y), n * d * h * w, c, 1 if act else 0, float(slope),
                                  _dt(x), _stream()), 'sg_bias_act_fwd')
        ctx.save_for_backward(y)
        ctx.act, ctx.slope, ctx.has_b = act, slope, b is not None
        ctx.b_ptr = b.data_ptr() if b is not None else 0
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        want_db = ctx.has_b and _wants(ctx, 1, ctx.b_ptr)
        # y only supplies the (piecewise constant) mask: detached, or a double backward would come back through this
        # node's own output with a materialised zero gradient and re-run the layer's whole backward for nothing
        g, gb = _BiasActBwd.apply(gy, y.detach() if ctx.act else None, ctx.slope, want_db, ctx.b_ptr)
        return g, (gb if want_db else None), None, None


class _PixelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        _note_all(x, eps)
        lib = _lib.load()
        _req_cuda(x)
        x = ndhwc(x)
        n, c, d, h, w = _dims(x)
        y = torch.empty_like(x)
        scale = torch.empty(n * d * h * w, device=x.device, dtype=torch.float32)
        check(lib.sg_pixel_norm_fwd(_ptr(x), _ptr(y), _ptr(scale), n * d * h * w, c, float(eps), _dt(x), _stream()),
              'sg_pixel_norm_fwd')
        ctx.save_for_backward(y, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        return _PixelNormBwd.apply(gy, y, scale), None


class _PixelNormBwd(torch.autograd.Function):
    """dx = scale * (dy - y * mean_c(dy*y)).  Only the generator uses pixel_norm and nothing differentiates
    its gradient a second time, so this node is once-differentiable."""

    @staticmethod
    def forward(ctx, gy, y, scale):
        _note_all(gy, y, scale)
        lib = _lib.load()
        gy, y = ndhwc(gy), ndhwc(y)
        n, c, d, h, w = _dims(gy)
        dx = torch.empty_like(gy)
        check(lib.sg_pixel_norm_bwd(_ptr(gy), _ptr(y), _ptr(scale), _ptr(dx), n * d * h * w, c, _dt(gy), _stream()),
              'sg_pixel_norm_bwd')
        return dx

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError('second-order gradient through pixel_norm is not part of the pgan step')


class _PnActBwd(torch.autograd.Function):
    """dz, db of y = pixel_norm(leaky_relu(z + b)) from dy in one pass (sg_pixel_norm_act_bwd); once-differentiable
    like _PixelNormBwd."""

    @staticmethod
    def forward(ctx, gy, y, scale, signs, slope, want_db, b_ptr=0):
        _note_all(gy, y, scale, signs, slope, want_db)
        lib = _lib.load()
        gy, y = ndhwc(gy), ndhwc(y)
        n, c, d, h, w = _dims(gy)
        nvox = n * d * h * w
        _check_signs(signs, nvox, c)
        dz = torch.empty_like(gy)
        db = ws = None
        if want_db:
            db = _f32_out(b_ptr, (c,), gy.device)
            ws = torch.empty(lib.sg_bias_act_bwd_workspace(c), device=gy.device, dtype=torch.uint8)
        check(lib.sg_pixel_norm_act_bwd(_ptr(gy), _ptr(y), _ptr(scale), _ptr(signs), float(slope), _ptr(dz), _ptr(db),
                                        _ptr(ws), nvox, c, _dt(gy), _stream()), 'sg_pixel_norm_act_bwd')
        if db is None:
            db = _zero_scalar(gy)
        ctx.mark_non_differentiable(db)
        return dz, db

    @staticmethod
    def backward(ctx, g, _gdb):
        raise NotImplementedError('second-order gradient through pixel_norm is not part of the pgan step')


class _Up(torch.autograd.Function):
    """y = gain * nearest-neighbour up-sampling of x by `factors` per dimension ((2,2,2): upscale3d / avg_unpool3d,
    networks/ops.py:250-262,276-289; (1,2,2): upscale2d of the 2-D tree), optionally times the LeakyReLU-backward mask
    given by `mask_bits`, the sign words of a tensor shaped like y (the gradient of downscale3d(leaky_relu(.)) in one
    pass)."""

    @staticmethod
    def forward(ctx, x, gain, mask_bits=None, mask_slope=0.0, factors=(2, 2, 2)):
        _note_all(x)
        lib = _lib.load()
        _req_cuda(x, mask_bits)
        x = ndhwc(x)
        n, c, d, h, w = _dims(x)
        fd, fh, fw = factors
        y = _empty_like_shape(x, c, (fd * d, fh * h, fw * w))
        _check_signs(mask_bits, fd * fh * fw * n * d * h * w, c)
        check(lib.sg_upscale_nn(_ptr(x), _ptr(y), _ptr(mask_bits), float(mask_slope), n, d, h, w, c, fd, fh, fw,
                                float(gain), _dt(x), _stream()), 'sg_upscale_nn')
        ctx.gain, ctx.mask_slope, ctx.factors = gain, mask_slope, tuple(factors)
        ctx.save_for_backward(mask_bits)
        return y

    @staticmethod
    def backward(ctx, gy):
        (mask_bits,) = ctx.saved_tensors
        # y = M * up(x): the gradient is down(M * gy), mask and block sum in one pass (sg_downscale_sum_masked)
        return _Down.apply(gy, ctx.gain, None, ctx.factors, mask_bits, ctx.mask_slope), None, None, None, None


class _Down(torch.autograd.Function):
    """y = gain * sum of each block of `factors` voxels (gain 1/8 with (2,2,2): downscale3d, networks/ops.py:265-273,
    292-305; gain 1/4 with (1,2,2): downscale2d)."""

    @staticmethod
    def forward(ctx, x, gain, in_info=None, factors=(2, 2, 2), mask_bits=None, mask_slope=0.0):
        _note_all(x)
        lib = _lib.load()
        _req_cuda(x)
        x = ndhwc(x)
        n, c, d, h, w = _dims(x)
        fd, fh, fw = factors
        y = _empty_like_shape(x, c, (d // fd, h // fh, w // fw))
        if mask_bits is not None:     # y = gain * block sum of M * x
            _check_signs(mask_bits, n * d * h * w, c)
            check(lib.sg_downscale_sum_masked(_ptr(x), _ptr(mask_bits), float(mask_slope), _ptr(y), n, d, h, w, c, fd, fh, fw,
                                              float(gain), _dt(x), _stream()), 'sg_downscale_sum_masked')
        else:
            check(lib.sg_downscale_sum(_ptr(x), _ptr(y), n, d, h, w, c, fd, fh, fw, float(gain), _dt(x), _stream()),
                  'sg_downscale_sum')
        ctx.gain, ctx.in_info, ctx.factors, ctx.mask_slope = gain, in_info, tuple(factors), mask_slope
        ctx.save_for_backward(mask_bits)
        return y

    @staticmethod
    def backward(ctx, gy):
        (mask_bits,) = ctx.saved_tensors
        if mask_bits is not None:     # adjoint of (block sum after mask) = mask after nearest up-scale
            return _Up.apply(gy, ctx.gain, mask_bits, ctx.mask_slope, ctx.factors), None, None, None, None, None
        if _masked_in(ctx.in_info):
            return _Up.apply(gy, ctx.gain, ctx.in_info.bits, ctx.in_info.slope, ctx.factors), None, None, None, None, None
        return _Up.apply(gy, ctx.gain, None, 0.0, ctx.factors), None, None, None, None, None


class _TriUp(torch.autograd.Function):
    """Trilinear x2 up-sampling, half-pixel centres (sg_trilinear_up2x); adjoint=True is its transpose (the gradient).
    The two are each other's backward, so the pair is differentiable to any order."""

    @staticmethod
    def forward(ctx, x, adjoint):
        _note_all(x, adjoint)
        lib = _lib.load()
        _req_cuda(x)
        x = ndhwc(x)
        n, c, d, h, w = _dims(x)
        if adjoint:
            if (d | h | w) & 1:
                raise ValueError('the adjoint takes a tensor of even extents')
            d, h, w = d // 2, h // 2, w // 2
        y = _empty_like_shape(x, c, (d, h, w) if adjoint else (2 * d, 2 * h, 2 * w))
        check(lib.sg_trilinear_up2x(_ptr(x), _ptr(y), n, d, h, w, c, 1 if adjoint else 0, _dt(x), _stream()),
              'sg_trilinear_up2x')
        ctx.adjoint = adjoint
        return y

    @staticmethod
    def backward(ctx, gy):
        return _TriUp.apply(gy, not ctx.adjoint), None


class DevCoef(float):
    """A step scalar that ALSO lives in device memory (element `idx` of `buf`, f32): the value a kernel of a captured step
    reads (sg_axpby_dev, sg_adam_ema_dev).  Behaves as its host value everywhere else."""

    def __new__(cls, value, buf, idx):
        o = float.__new__(cls, value)
        o.buf, o.idx = buf, int(idx)
        return o

    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 4 * self.idx)


class DevScalars:
    """The per-step scalars of a captured step (optimization.StepGraph): fade-in weights [alpha, 1 - alpha] and the step size
    of each optimiser.  The host writes them into a pinned mirror (`set`) and `flush()` sends all of them with one
    asynchronous copy before the graph is replayed; the captured kernels read the device copy.  The values are the f32
    roundings of the doubles the eager path hands to ctypes: the same bits reach the same arithmetic.
    A replayed step costs the host next to nothing, so it runs ahead of the device: the pinned source of a copy that has not
    executed yet must not be rewritten (with ONE mirror a replay could read the alpha / step size of a later
    step).  The mirror is a ring of pinned slots, each with the event recorded behind its last copy; a slot is rewritten only
    after that event (the host waits only if it is RING flushes ahead)."""
    RING = 8

    def __init__(self, device, n=8):
        cuda = torch.device(device).type == 'cuda'
        self._ring = [torch.zeros(n, dtype=torch.float32).pin_memory() if cuda else torch.zeros(n, dtype=torch.float32)
                      for _ in range(self.RING if cuda else 1)]
        self._events = [None] * len(self._ring)
        self._cur = 0
        self.host = self._ring[0]
        self.dev = torch.zeros(n, dtype=torch.float32, device=device)
        self.waits = 0            # times the host had to wait for a slot (tests)

    def set(self, idx, value):
        self.host[idx] = float(value)

    def coef(self, idx, value):
        self.set(idx, value)
        return DevCoef(value, self.dev, idx)

    def flush(self):
        self.dev.copy_(self.host, non_blocking=True)
        if len(self._ring) == 1:
            return
        ev = self._events[self._cur]
        if ev is None:
            ev = self._events[self._cur] = torch.cuda.Event()
        ev.record()
        nxt = (self._cur + 1) % len(self._ring)
        if self._events[nxt] is not None and not self._events[nxt].query():
            self._events[nxt].synchronize()
            self.waits += 1
        self._ring[nxt].copy_(self.host)      # values not set again keep theirs
        self._cur, self.host = nxt, self._ring[nxt]


class _Axpby(torch.autograd.Function):
    """out = wa*a + wb*b (fade-in lerp, pgan/generator.py:100-101, pgan/discriminator.py:105)."""

    @staticmethod
    def forward(ctx, a, b, wa, wb):
        _note_all(a, b, wa, wb)
        lib = _lib.load()
        _req_cuda(a, b)
        a = ndhwc(a)
        if b is not None:
            b = ndhwc(b)
            if b.shape != a.shape or b.dtype != a.dtype:
                raise ValueError('lerp operands differ in shape or dtype')
        out = torch.empty_like(a)
        if isinstance(wa, DevCoef):      # captured step: [wa, wb] are adjacent device floats (wb unread without b)
            if b is not None and not (isinstance(wb, DevCoef) and wb.buf is wa.buf and wb.idx == wa.idx + 1):
                raise ValueError('device-side lerp weights must be adjacent elements of one DevScalars buffer')
            check(lib.sg_axpby_dev(_ptr(a), _ptr(b), _ptr(out), wa.ptr(), a.numel(), _dt(a), _stream()), 'sg_axpby_dev')
        else:
            check(lib.sg_axpby(_ptr(a), _ptr(b), _ptr(out), float(wa), float(wb), a.numel(), _dt(a), _stream()),
                  'sg_axpby')
        ctx.wa, ctx.wb, ctx.has_b = wa, wb, b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        def scaled(w):      # w * g; a weight of exactly 1 (the stabilising phase's alpha = 0 side) is the identity: no pass
            return g if float(w) == 1.0 else _Axpby.apply(g, None, w, 0.0)
        ga = scaled(ctx.wa) if ctx.needs_input_grad[0] else None
        gb = scaled(ctx.wb) if (ctx.has_b and ctx.needs_input_grad[1]) else None
        return ga, gb, None, None


class _AddNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stddev, seed, offset):
        """offset: a Python int, or a one-element int64 DEVICE tensor holding the Philox offset (read by the kernel and then
        advanced by 1 << 40 on the device: the form a captured step uses, sg_add_noise_dev)."""
        _note_all(x, stddev, seed, offset)
        lib = _lib.load()
        _req_cuda(x)
        x = ndhwc(x)
        out = torch.empty_like(x)
        if torch.is_tensor(offset):
            check(lib.sg_add_noise_dev(_ptr(x), _ptr(out), float(stddev), int(seed), _ptr(offset), 1 << 40, x.numel(), _dt(x),
                                       _stream()), 'sg_add_noise_dev')
        else:
            check(lib.sg_add_noise(_ptr(x), _ptr(out), float(stddev), int(seed), int(offset), x.numel(), _dt(x),
                                   _stream()), 'sg_add_noise')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


class _SumsqKeepW(torch.autograd.Function):
    """out[n, w] = sum_{c,d,h} g^2: tf.reduce_sum(tf.square(g), (1,2,3)) on NCDHW (networks/loss.py:140)."""

    @staticmethod
    def forward(ctx, g):
        _note_all(g)
        lib = _lib.load()
        _req_cuda(g)
        g = ndhwc(g)
        n, c, d, h, w = _dims(g)
        out = torch.empty((n, w), device=g.device, dtype=torch.float32)
        check(lib.sg_sumsq_ndhwc_keep_w(_ptr(g), _ptr(out), n, d, h, w, c, _dt(g), _stream()),
              'sg_sumsq_ndhwc_keep_w')
        ctx.save_for_backward(g)
        return out

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        n, w = gout.shape
        return (2.0 * gout.reshape(n, 1, 1, 1, w)).to(g.dtype) * g


# ---------------------------------------------------------------------------------------------------
# public functional API
# ---------------------------------------------------------------------------------------------------
PN_FUSE_MAX_CHANNELS = int(os.environ.get('SARAGAN_PN_FUSE_MAX', '64'))   # widest layer whose pixel_norm rides in the conv epilogue


def conv3d(x, w, coef=1.0, bias=None, act=False, slope=0.2, pixel_norm=False, eps=1e-8, upsample_in=False,
           fuse=True, out_info=None, in_info=None):
    """conv3d (+ optional fused nearest-x2 of the input, bias, LeakyReLU, pixel-norm)."""
    if bias is None and not act and not pixel_norm:
        return _Conv.apply(x, w, coef, False, upsample_in, in_info)
    cout = w.shape[-1]
    if fuse and (not pixel_norm or cout <= 128):
        return _ConvBiasAct.apply(x, w, bias, coef, upsample_in, act, slope, pixel_norm, eps, out_info, in_info)
    y = _Conv.apply(x, w, coef, False, upsample_in, in_info)
    y = _BiasAct.apply(y, bias, act, slope)
    return _PixelNorm.apply(y, eps) if pixel_norm else y


def conv3d_pn_to_rgb(x, w, coef, bias, ups, slope, eps, in_info, w_rgb, coef_rgb, bias_rgb):
    """(img, y): y = pixel_norm(leaky_relu(conv3d(x) + bias)), img = conv3d(y, w_rgb) + bias_rgb, one autograd node."""
    return _ConvPnActToRgb.apply(x, w, bias, coef, ups, slope, eps, in_info, w_rgb, bias_rgb, coef_rgb)


def conv3d_act_pool(x, w, coef, bias, slope, in_info=None):
    return _ConvBiasActPool.apply(x, w, bias, coef, slope, in_info)


def bias_act(x, bias, act=False, slope=0.2):
    return _BiasAct.apply(x, bias, act, slope)


def pixel_norm(x, eps=1e-8):
    return _PixelNorm.apply(x, eps)


def upscale2x(x, gain=1.0, factors=(2, 2, 2)):
    return _Up.apply(x, gain, None, 0.0, factors)


def downscale2x(x, gain=0.125, in_info=None, factors=(2, 2, 2)):
    return _Down.apply(x, gain, in_info, factors)


def upscale_trilinear2x(x):
    return _TriUp.apply(x, False)


def lerp(a, b, wa, wb):
    return _Axpby.apply(a, b, wa, wb)


def interpolate_rows(gamma, a, b):
    """gamma * a + (1 - gamma) * b with one weight per batch sample (gamma: [N, 1, 1, 1, 1] or [N], f32): the gradient
    penalty's interpolates (networks/loss.py:70-71, :133-134) in ONE launch with one rounding (sg_lerp_rows).  No gradient:
    every caller detaches the result (the penalty differentiates with respect to the interpolates, not through them)."""
    lib = _lib.load()
    _req_cuda(a, b, gamma)
    a, b = ndhwc(a.detach()), ndhwc(b.detach())
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError('interpolate_rows: operands differ in shape or dtype')
    g = gamma.detach().reshape(-1).float().contiguous()
    if g.numel() != a.shape[0]:
        raise ValueError('interpolate_rows: one weight per batch sample')
    out = torch.empty_like(a)
    check(lib.sg_lerp_rows(_ptr(a), _ptr(b), _ptr(g), _ptr(out), a.shape[0], a.numel() // a.shape[0], _dt(a), _stream()),
          'sg_lerp_rows')
    return out


def add_noise(x, stddev, seed, offset=0):
    return _AddNoise.apply(x, stddev, seed, offset)


def sumsq_keep_w(g):
    return _SumsqKeepW.apply(g)


class _MinibatchStddev(torch.autograd.Function):
    """networks/ops.py:313-325: concat(x, per-group mean stddev).  Once-differentiable (the layer is disabled in
    pgan, pgan/discriminator.py:50, so it never sits under the gradient penalty)."""

    @staticmethod
    def forward(ctx, x, group_size):
        _note_all(x, group_size)
        lib = _lib.load()
        _req_cuda(x)
        x = ndhwc(x)
        n, c, d, h, w = _dims(x)
        g = min(group_size, n)
        if n % g != 0:
            raise ValueError(f'minibatch_stddev_layer: batch {n} not divisible by group {g}')
        y = _empty_like_shape(x, c + 1)
        ws = torch.empty(n // g, device=x.device, dtype=torch.float32)
        check(lib.sg_minibatch_stddev_fwd(_ptr(x), _ptr(y), _ptr(ws), n, d * h * w, c, group_size, _dt(x), _stream()),
              'sg_minibatch_stddev_fwd')
        ctx.save_for_backward(x)
        ctx.group_size = group_size
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        gy = ndhwc(gy)
        n, c, d, h, w = _dims(x)
        g = min(ctx.group_size, n)
        dx = torch.empty_like(x)
        ws = torch.empty(n // g, device=x.device, dtype=torch.float32)
        check(lib.sg_minibatch_stddev_bwd(_ptr(gy), _ptr(x), _ptr(dx), _ptr(ws), n, d * h * w, c, ctx.group_size, _dt(x),
                                          _stream()), 'sg_minibatch_stddev_bwd')
        return dx, None


def minibatch_stddev(x, group_size=4):
    """networks/ops.py:313-325 (the layer is disabled in pgan, pgan/discriminator.py:50)."""
    return _MinibatchStddev.apply(x, group_size)


def adam_step_size(lr, beta1, beta2, step):
    """lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) (tf.train.AdamOptimizer), in host doubles."""
    return lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)


def adam_ema_(p, g, m, v, ema, lr, beta1, beta2, step, eps=1e-8, gscale=1.0, ema_decay=0.99, lr_dev=None):
    """In-place fused TF-Adam + EMA over flat f32 buffers.  lr_dev: a DevCoef holding lr_t on the
    device (captured step): `lr` and `step` are then not used."""
    lib = _lib.load()
    _req_cuda(p, g, m, v, ema)
    if g is not None:
        mark_packs_stale()   # the kernel rewrites parameters behind torch's version counters
    if lr_dev is not None and g is not None:
        check(lib.sg_adam_ema_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), p.numel(), lr_dev.ptr(), float(beta1),
                                  float(beta2), float(eps), float(gscale), float(ema_decay), _stream()), 'sg_adam_ema_dev')
        return
    lr_t = adam_step_size(lr, beta1, beta2, step) if g is not None else 0.0
    check(lib.sg_adam_ema(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), p.numel(), float(lr_t), float(beta1),
                          float(beta2), float(eps), float(gscale), float(ema_decay), _stream()), 'sg_adam_ema')


def optim_step_(kind, p, g, s1, s2, ema, lr, h=0.0, eps=0.0, nesterov=False, gscale=1.0, ema_decay=0.99, lr_dev=None):
    """In-place fused SGD / Momentum / Adadelta (+ EMA) over flat f32 buffers (sg_optim_step; lr_dev: the learning rate as
    a DevCoef on the device, captured step)."""
    lib = _lib.load()
    _req_cuda(p, g, s1, s2, ema)
    mark_packs_stale()      # the kernel rewrites parameters behind torch's version counters
    if lr_dev is not None:
        check(lib.sg_optim_step_dev(int(kind), _ptr(p), _ptr(g), _ptr(s1), _ptr(s2), _ptr(ema), p.numel(), lr_dev.ptr(), float(h),
                                    float(eps), 1 if nesterov else 0, float(gscale), float(ema_decay), _stream()),
              'sg_optim_step_dev')
        return
    check(lib.sg_optim_step(int(kind), _ptr(p), _ptr(g), _ptr(s1), _ptr(s2), _ptr(ema), p.numel(), float(lr), float(h),
                            float(eps), 1 if nesterov else 0, float(gscale), float(ema_decay), _stream()),
          'sg_optim_step')


def segment_sumsq(flat, offsets_dev, nseg):
    lib = _lib.load()
    out = torch.empty(nseg, device=flat.device, dtype=torch.float32)
    check(lib.sg_segment_sumsq(_ptr(flat), _ptr(offsets_dev), _ptr(out), nseg, _stream()), 'sg_segment_sumsq')
    return out
