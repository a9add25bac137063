"""HBM roofline of the non-convolution kernels at their in-step shapes (pgan 's' phase 6, batch 32: the tensors of the
128x128x32 and 64x64x16 levels): algorithmic bytes (reads + writes) / measured time, against the 6.3 TB/s a float4 copy
achieves on MI355X (MI355X_MICROARCH.md).  usage: python tools/ew_roofline.py [--dtype bf16] [--iters 20] [--json out]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib  # noqa: E402

ACHIEVABLE = 6300.0   # GB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    lib = _lib.load()
    dt = _lib.SG_BF16 if a.dtype == 'bf16' else _lib.SG_F32
    tdt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    es = 2 if a.dtype == 'bf16' else 4
    dev = torch.device('cuda:0')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []

    def run(name, shape_note, nbytes, fn):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        gbs = nbytes / ms / 1e6
        rows.append(dict(kernel=name, shape=shape_note, bytes=int(nbytes), ms=round(ms, 4), GBps=round(gbs, 1),
                         frac_of_achievable=round(gbs / ACHIEVABLE, 3)))
        print(f'{name:28s} {shape_note:34s} {nbytes / 1e9:7.3f} GB {ms:8.3f} ms {gbs:8.1f} GB/s  {gbs / ACHIEVABLE:5.2f}', flush=True)

    def t(*shape, dtype=None):
        return torch.randn(*shape, device=dev).to(dtype or tdt)

    for (n, d, h, w, c) in ((64, 32, 128, 128, 64), (32, 32, 128, 128, 32), (64, 16, 64, 64, 128)):
        nvox = n * d * h * w
        note = f'n{n} {d}x{h}x{w} c{c}'
        x = t(n, d, h, w, c)
        y = torch.randn_like(x)
        z = torch.empty_like(x)
        nw = (c + 31) // 32
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int32)
        half = torch.empty(n, d // 2, h // 2, w // 2, c, device=dev, dtype=tdt)
        bias = torch.zeros(c, device=dev)
        db = torch.empty(c, device=dev)
        ws = torch.empty(lib.sg_bias_act_bwd_workspace(c), device=dev, dtype=torch.uint8)
        scale = torch.rand(nvox, device=dev) + 0.5
        run('downscale2x', note, nvox * c * es * (1 + 1 / 8),
            lambda: _lib.check(lib.sg_downscale2x(x.data_ptr(), half.data_ptr(), n, d, h, w, c, 0.125, dt, st)))
        run('upscale2x_masked', note, nvox * c * es * (1 + 1 / 8) + nvox * nw * 4,
            lambda: _lib.check(lib.sg_upscale2x_masked(half.data_ptr(), z.data_ptr(), bits.data_ptr(), 0.2, n, d // 2, h // 2, w // 2, c, 0.125, dt, st)))
        run('bias_act_bwd_bits (+db)', note, nvox * c * es * 2 + nvox * nw * 4,
            lambda: _lib.check(lib.sg_bias_act_bwd_bits(x.data_ptr(), bits.data_ptr(), z.data_ptr(), db.data_ptr(), ws.data_ptr(), nvox, c, 0.2, dt, st)))
        run('bias_act_bwd (y)', note, nvox * c * es * 3,
            lambda: _lib.check(lib.sg_bias_act_bwd(x.data_ptr(), y.data_ptr(), z.data_ptr(), None, ws.data_ptr(), nvox, c, 0.2, dt, st)))
        run('pixel_norm_act_bwd', note, nvox * c * es * 3 + nvox * (4 + nw * 4),
            lambda: _lib.check(lib.sg_pixel_norm_act_bwd(x.data_ptr(), y.data_ptr(), scale.data_ptr(), bits.data_ptr(), 0.2, z.data_ptr(), db.data_ptr(), ws.data_ptr(), nvox, c, dt, st)))
        run('axpby (lerp)', note, nvox * c * es * 3,
            lambda: _lib.check(lib.sg_axpby(x.data_ptr(), y.data_ptr(), z.data_ptr(), 0.3, 0.7, nvox * c, dt, st)))
        run('sign_words', note, nvox * c * es + nvox * nw * 4,
            lambda: _lib.check(lib.sg_sign_words(x.data_ptr(), bits.data_ptr(), nvox, c, dt, st)))
        del x, y, z, bits, half
    # 1x1x1 rgb layers at the top level (from_rgb 1 -> 32, to_rgb 32 -> 1) through sg_conv3d_fwd
    from saragan_amd._lib import ConvEpilogue, ConvShape
    for (n, cin, cout) in ((96, 1, 32), (32, 32, 1)):
        d, h, w = 32, 128, 128
        nvox = n * d * h * w
        shp = ConvShape(n, d, h, w, cin, cout, 1, 1, 1, 0)
        x = t(n, d, h, w, cin)
        y = torch.empty(n, d, h, w, cout, device=dev, dtype=tdt)
        wt = torch.randn(1, 1, 1, cin, cout, device=dev)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 1.0, 0, wp.data_ptr(), C.byref(shp), dt, st))
        bias = torch.zeros(cout, device=dev)
        sout = torch.empty(n, d, h, w, (cout + 31) // 32, device=dev, dtype=torch.int32)
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
        run(f'pointwise conv {cin}->{cout}', f'n{n} {d}x{h}x{w}', nvox * (cin + cout) * es + nvox * 4,
            lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st)))
    # fused Adam + EMA over 29 M parameters (one network of pgan 's')
    npar = 28_900_000 // 4 * 4
    p, g_, m, v, e = (torch.randn(npar, device=dev) for _ in range(5))
    v.abs_()
    run('adam_ema', f'{npar / 1e6:.1f} M params', npar * 36,
        lambda: _lib.check(lib.sg_adam_ema(p.data_ptr(), g_.data_ptr(), m.data_ptr(), v.data_ptr(), e.data_ptr(), npar, 1e-3, 0.0, 0.9, 1e-8, 1.0, 0.99, st)))
    if a.json:
        json.dump(dict(dtype=a.dtype, achievable_GBps=ACHIEVABLE, rows=rows), open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
