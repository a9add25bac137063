"""Launches the hot kernels of the pgan 's' phase-6 step (bench default: batch 32 per GPU) at their in-step shapes, each
twice, after a streaming kernel of known byte count (calibration), for the rocprofv3 counter passes -- one counter
group per pass, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass):

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_rd -o rd -- python3 tools/pmc_probe.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_wr -o wr -- python3 tools/pmc_probe.py
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
            -d gpurun_out/pmc_mfma -o mf -- python3 tools/pmc_probe.py

The probe writes gpurun_out/pmc_manifest.json: one record per measured launch (kernel-name substring to look for,
algorithmic bytes, FLOPs); tools/pmc_summary.py joins it with the counter CSVs into profiles/r02_pmc_*.json."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402

CONVS = [  # n, (d,h,w), cin, cout (all 3x3x3): the >= 1 ms/step entries of `bench.py --dump-prof`; n = 64 is the
    # concatenated real+fake batch of the discriminator, n = 32 the generator / gradient-penalty passes
    (64, (32, 128, 128), 64, 32),
    (64, (32, 128, 128), 32, 32),
    (64, (32, 128, 128), 32, 64),
    (32, (32, 128, 128), 32, 64),
    (32, (32, 128, 128), 32, 32),
    (32, (32, 128, 128), 64, 32),
    (64, (16, 64, 64), 64, 64),
    (64, (16, 64, 64), 64, 128),
    (64, (16, 64, 64), 128, 64),
    (64, (8, 32, 32), 128, 128),
    (32, (4, 16, 16), 128, 128),       # round 4: the 16-wide level (conv_wgrad3l<w16>)
    (32, (4, 16, 16), 128, 512),
]


def main():
    lib = _lib.load()
    # round 4: the interleaved 64 -> 32 layers run ONE pass (csrc/conv3p.hip) unless SG_FWD_NO_3P=1 keeps the two-pass K split
    one_pass = os.environ.get('SG_FWD_NO_3P', '0') != '1'
    dt = _lib.SG_BF16
    dev = torch.device('cuda:0')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    manifest = []

    def twice(tag, match, alg_bytes, flops, fn, per_call=1):
        for rep in range(2):
            fn()
        torch.cuda.synchronize()
        manifest.append(dict(tag=tag, match=match, algorithmic_bytes=int(alg_bytes), flops=float(flops), repeats=2 * per_call,
                             dispatches_per_call=per_call))
        print(tag, 'done', flush=True)

    # calibration: out = 1*a + 0.5*b over 2^28 bf16 elements: reads 2 x 512 MiB, writes 512 MiB, 16 B per lane
    numel = 1 << 28
    a = torch.randn(numel, device=dev).to(torch.bfloat16)
    b = torch.randn(numel, device=dev).to(torch.bfloat16)
    o = torch.empty_like(a)
    twice('calibration axpby 2^28', 'axpby', 3 * numel * 2, 0,
          lambda: _lib.check(lib.sg_axpby(a.data_ptr(), b.data_ptr(), o.data_ptr(), 1.0, 0.5, numel, dt, st)))
    del a, b, o
    for n, (d, h, w), cin, cout in CONVS:
        shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
        vox = n * d * h * w
        x = torch.randn(n, d, h, w, cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(n, d, h, w, cout, device=dev).to(torch.bfloat16)
        wt = torch.randn(3, 3, 3, cin, cout, device=dev)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
        y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
        bias = torch.zeros(cout, device=dev)
        nw = (cout + 31) // 32
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int32)
        sout = torch.empty_like(bits)
        ep_plain = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
        ep_mask = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None)
        # as functional.raw_conv does: scratch for the two-pass (K-split) path where the library has one for this shape
        fws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
        fws = torch.empty(max(16, fws_bytes), device=dev, dtype=torch.uint8)
        per_call = 2 if fws_bytes else 1
        if fws_bytes and cin == 64 and one_pass:
            per_call = 1
        if fws_bytes:
            for e_ in (ep_plain, ep_mask):
                e_.workspace, e_.workspace_bytes = fws.data_ptr(), fws_bytes
        wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        dw = torch.empty(3, 3, 3, cin, cout, device=dev)
        db = torch.empty(cout, device=dev)
        flops = 2.0 * vox * cin * cout * 27
        name = f'n{n} {d}x{h}x{w} {cin}->{cout}'
        twice(f'fwd bias+lrelu+sign_out {name}', 'conv_fwd', vox * (cin + cout) * 2 + vox * nw * 4 + 27 * cin * cout * 2, flops,
              lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep_plain), dt, st)), per_call)
        twice(f'fwd mask_bits {name}', 'conv_fwd', vox * (cin + cout) * 2 + vox * nw * 4 + 27 * cin * cout * 2, flops,
              lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep_mask), dt, st)), per_call)
        twice(f'wgrad+dbias {name}', 'conv_wgrad', vox * (cin + cout) * 2 + 27 * cin * cout * 4, flops,
              lambda: _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), 1.0,
                                                          ws.data_ptr(), wsb, C.byref(shp), dt, st)))
        if fws_bytes and cin == 64:   # the same layer reading x as two 32-channel tensors (sg_conv_epilogue.x_plane_channels)
            ep_pl = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None)
            ep_pl.workspace, ep_pl.workspace_bytes, ep_pl.x_plane_channels = fws.data_ptr(), fws_bytes, 32
            twice(f'fwd mask_bits, x as two 32-channel tensors {name}', 'conv_fwd', vox * (cin + cout) * 2 + vox * nw * 4 + 27 * cin * cout * 2, flops,
                  lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep_pl), dt, st)), 2)
        if cin <= 32 and cout % 32 == 0:      # the fused first stage of downscale3d (sg_conv_epilogue.pool)
            yp = torch.empty(n, d // 2, h, w // 2, cout, device=dev, dtype=torch.bfloat16)
            ep_pool = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
            ep_pool.pool = 1
            twice(f'fwd pooled (D x W mean) {name}', 'conv_fwd3', vox * (cin + cout / 4) * 2 + vox * nw * 4 + 27 * cin * cout * 2, flops,
                  lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), yp.data_ptr(), C.byref(shp), C.byref(ep_pool), dt, st)))
        del x, dy, y, bits, sout, ws, fws
    # round 3: the sub-pixel up-convolution (one launch for all eight classes), low-resolution input -> fine output
    for n, (d, h, w), cin, cout, pn in ((32, (16, 64, 64), 64, 32, 1), (32, (8, 32, 32), 128, 64, 1), (32, (4, 16, 16), 128, 128, 0)):
        shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
        if not lib.sg_upconv3d_subpixel_supported(C.byref(shp), dt):
            continue
        vox = n * d * h * w
        x = torch.randn(n, d, h, w, cin, device=dev).to(torch.bfloat16)
        wt = torch.randn(3, 3, 3, cin, cout, device=dev)
        wp = torch.empty(lib.sg_upconv3d_subpixel_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_upconv3d_subpixel_pack(wt.data_ptr(), 0.05, wp.data_ptr(), C.byref(shp), dt, st))
        y = torch.empty(n, 2 * d, 2 * h, 2 * w, cout, device=dev, dtype=torch.bfloat16)
        nw = (cout + 31) // 32
        sout = torch.empty(n, 2 * d, 2 * h, 2 * w, nw, device=dev, dtype=torch.int32)
        scale = torch.empty(8 * vox, device=dev)
        bias = torch.zeros(cout, device=dev)
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, pn, 1e-8, scale.data_ptr() if pn else None, None, 0.0, sout.data_ptr())
        # algorithmic bytes: the low-resolution input once, the fine output, its sign words (+ scale), the summed weights
        alg = vox * cin * 2 + 8 * vox * (cout * 2 + nw * 4 + (4 if pn else 0)) + 64 * cin * cout * 2
        twice(f'upconv sub-pixel fwd bias+lrelu{"+pn" if pn else ""}+sign_out n{n} {2 * d}x{2 * h}x{2 * w} {cin}->{cout}', 'upconv_subpixel_fwd', alg,
              2.0 * vox * 64 * cin * cout,
              lambda: _lib.check(lib.sg_upconv3d_subpixel_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st)))
        del x, y, sout, scale
    # round 3: the sub-pixel data gradient (fine gradient in, low-resolution gradient out)
    for n, (d, h, w), ci, co in ((32, (16, 64, 64), 64, 32), (32, (8, 32, 32), 128, 64), (32, (4, 16, 16), 128, 128)):
        shp = ConvShape(n, d, h, w, ci, co, 3, 3, 3, 0)
        if not lib.sg_upconv3d_subpixel_dgrad_supported(C.byref(shp), dt):
            continue
        vox = n * d * h * w
        gy = torch.randn(n, 2 * d, 2 * h, 2 * w, co, device=dev).to(torch.bfloat16)
        wt = torch.randn(3, 3, 3, ci, co, device=dev)
        wp = torch.empty(lib.sg_upconv3d_subpixel_dgrad_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_upconv3d_subpixel_dgrad_pack(wt.data_ptr(), 0.05, wp.data_ptr(), C.byref(shp), dt, st))
        gx = torch.empty(n, d, h, w, ci, device=dev, dtype=torch.bfloat16)
        twice(f'upconv sub-pixel dgrad n{n} {2 * d}x{2 * h}x{2 * w} {co}->{ci}', 'upconv_subpixel_dgrad_kernel', 8 * vox * co * 2 + vox * ci * 2 + 64 * ci * co * 2,
              2.0 * vox * 64 * ci * co,
              lambda: _lib.check(lib.sg_upconv3d_subpixel_dgrad(gy.data_ptr(), wp.data_ptr(), gx.data_ptr(), C.byref(shp), dt, st)))
        del gy, gx
    # round 3: D's pooled backward with the masked gather fused into its consumers (pooled gradient + sign words in)
    for n in (32, 64):
        d, h, w = 32, 128, 128
        vox = n * d * h * w
        gyh = torch.randn(n, d // 2, h // 2, w // 2, 64, device=dev).to(torch.bfloat16)
        x = torch.randn(n, d, h, w, 32, device=dev).to(torch.bfloat16)
        bits64 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 2), device=dev, dtype=torch.int32)
        bits32 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 1), device=dev, dtype=torch.int32)
        wt = torch.randn(3, 3, 3, 32, 64, device=dev)
        shp = ConvShape(n, d, h, w, 64, 32, 3, 3, 3, 1)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 1, wp.data_ptr(), C.byref(shp), dt, st))
        gx = torch.empty(n, d, h, w, 32, device=dev, dtype=torch.bfloat16)
        fws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
        fws = torch.empty(max(16, fws_bytes), device=dev, dtype=torch.uint8)
        ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, bits32.data_ptr(), 0.2, None)
        ep.workspace, ep.workspace_bytes = fws.data_ptr(), fws_bytes
        ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = bits64.data_ptr(), 0.2, 0.125
        # algorithmic bytes: pooled gradient once, the input's sign words, output + its mask words, weights
        alg = vox / 8 * 64 * 2 + vox * 2 * 4 + vox * 32 * 2 + vox * 4 + 27 * 64 * 32 * 2
        twice(f'fwd masked gather ({"one pass" if one_pass else "K split"}) n{n} {d}x{h}x{w} 64->32', 'conv_fwd3p' if one_pass else 'conv_fwd3s', alg,
              2.0 * vox * 64 * 32 * 27,
              lambda: _lib.check(lib.sg_conv3d_fwd(gyh.data_ptr(), wp.data_ptr(), gx.data_ptr(), C.byref(shp), C.byref(ep), dt, st)),
              1 if one_pass else 2)
        shw = ConvShape(n, d, h, w, 32, 64, 3, 3, 3, 0)
        wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shw), dt)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        dw = torch.empty(3, 3, 3, 32, 64, device=dev)
        db = torch.empty(64, device=dev)
        twice(f'wgrad+dbias, dy gathered n{n} {d}x{h}x{w} 32->64', 'conv_wgrad3l', vox * 32 * 2 + vox / 8 * 64 * 2 + vox * 2 * 4, 2.0 * vox * 64 * 32 * 27,
              lambda: _lib.check(lib.sg_conv3d_wgrad_bias_up_masked(x.data_ptr(), gyh.data_ptr(), bits64.data_ptr(), 0.2, 0.125, dw.data_ptr(),
                                                                    db.data_ptr(), 0.05, ws.data_ptr(), wsb, C.byref(shw), dt, st)))
        del gyh, x, bits64, bits32, gx, fws, ws
    # round 3: GEMM-tiled convolution of the 1x4x4 / 2x8x8 levels (K split: the partial tiles are part of the traffic)
    for n, (d, h, w), cin, cout in ((32, (2, 8, 8), 512, 512), (64, (2, 8, 8), 512, 512), (32, (1, 4, 4), 512, 512)):
        shp = ConvShape(n, d, h, w, cin, cout, 1, 3, 3, 0)
        vox = n * d * h * w
        x = torch.randn(n, d, h, w, cin, device=dev).to(torch.bfloat16)
        wt = torch.randn(1, 3, 3, cin, cout, device=dev)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
        y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
        sout = torch.empty(n, d, h, w, cout // 32, device=dev, dtype=torch.int32)
        bias = torch.zeros(cout, device=dev)
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
        fws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
        fws = torch.empty(max(16, fws_bytes), device=dev, dtype=torch.uint8)
        if fws_bytes:
            ep.workspace, ep.workspace_bytes = fws.data_ptr(), fws_bytes
        twice(f'gemm conv fwd bias+lrelu+sign_out n{n} {d}x{h}x{w} {cin}->{cout}', 'conv_gemm', vox * (cin + cout) * 2 + 9 * cin * cout * 2,
              2.0 * vox * cin * cout * 9,
              lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st)), 2 if fws_bytes else 1)
        del x, y, sout, fws
    # round 3: small-channel 2-D kernels (BASELINE config 5's top level, fp32)
    f32 = _lib.SG_F32
    for n, hw, cin, cout in ((8, 1024, 4, 8), (8, 1024, 4, 4), (4, 1024, 8, 4), (8, 512, 8, 16), (4, 512, 16, 8)):
        shp = ConvShape(n, 1, hw, hw, cin, cout, 1, 3, 3, 0)
        vox = n * hw * hw
        x = torch.randn(n, 1, hw, hw, cin, device=dev)
        dy = torch.randn(n, 1, hw, hw, cout, device=dev)
        wt = torch.randn(1, 3, 3, cin, cout, device=dev)
        wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), f32), device=dev, dtype=torch.uint8)
        _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), f32, st))
        y = torch.empty(n, 1, hw, hw, cout, device=dev)
        sout = torch.empty(n, 1, hw, hw, 1, device=dev, dtype=torch.int32)
        bias = torch.zeros(cout, device=dev)
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
        wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), f32)
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        dw = torch.empty(1, 3, 3, cin, cout, device=dev)
        db = torch.empty(cout, device=dev)
        name = f'f32 n{n} {hw}x{hw} {cin}->{cout}'
        twice(f'small fwd bias+lrelu+sign_out {name}', 'conv_small_fwd', vox * (cin + cout) * 4 + vox * 4, 2.0 * vox * cin * cout * 9,
              lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), f32, st)))
        twice(f'small wgrad+dbias {name}', 'conv_small_wgrad_kernel', vox * (cin + cout) * 4, 2.0 * vox * cin * cout * 9,
              lambda: _lib.check(lib.sg_conv3d_wgrad_bias(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), 1.0,
                                                          ws.data_ptr(), wsb, C.byref(shp), f32, st)))
        del x, dy, y, sout, ws
    # non-convolution kernels at the top level
    for (n, d, h, w, c) in ((64, 32, 128, 128, 64), (32, 32, 128, 128, 32)):
        vox = n * d * h * w
        nw = (c + 31) // 32
        x = torch.randn(n, d, h, w, c, device=dev).to(torch.bfloat16)
        y = torch.randn_like(x)
        z = torch.empty_like(x)
        half = torch.empty(n, d // 2, h // 2, w // 2, c, device=dev, dtype=torch.bfloat16)
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int32)
        db = torch.empty(c, device=dev)
        ws = torch.empty(lib.sg_bias_act_bwd_workspace(c), device=dev, dtype=torch.uint8)
        scale = torch.rand(vox, device=dev) + 0.5
        name = f'n{n} {d}x{h}x{w} c{c}'
        twice(f'downscale2x {name}', 'downscale2x', vox * c * 2 * (1 + 1 / 8), 0,
              lambda: _lib.check(lib.sg_downscale2x(x.data_ptr(), half.data_ptr(), n, d, h, w, c, 0.125, dt, st)))
        twice(f'upscale2x_masked {name}', 'upscale2x', vox * c * 2 * (1 + 1 / 8) + vox * nw * 4, 0,
              lambda: _lib.check(lib.sg_upscale2x_masked(half.data_ptr(), z.data_ptr(), bits.data_ptr(), 0.2, n, d // 2, h // 2, w // 2, c, 0.125, dt, st)))
        twice(f'bias_act_bwd_bits+db {name}', 'bias_act_bwd', vox * c * 2 * 2 + vox * nw * 4, 0,
              lambda: _lib.check(lib.sg_bias_act_bwd_bits(x.data_ptr(), bits.data_ptr(), z.data_ptr(), db.data_ptr(), ws.data_ptr(), vox, c, 0.2, dt, st)))
        twice(f'pixel_norm_act_bwd {name}', 'pixel_norm', vox * c * 2 * 3 + vox * (4 + nw * 4), 0,
              lambda: _lib.check(lib.sg_pixel_norm_act_bwd(x.data_ptr(), y.data_ptr(), scale.data_ptr(), bits.data_ptr(), 0.2, z.data_ptr(), db.data_ptr(), ws.data_ptr(), vox, c, dt, st)))
        del x, y, z, half, bits
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    json.dump(manifest, open(os.path.join(out, 'pmc_manifest.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
