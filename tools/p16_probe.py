"""Diagnostic: the one-pass 64 -> 32 kernel on v_mfma_f32_32x32x16_bf16 (default) against its v_mfma_f32_16x16x32_bf16 variant
(SG_FWD3P_16=1), A/B in one process at the bench shape: plain interleaved input (bias + LeakyReLU + sign words; output mask) and the
fused masked gather.  usage: python tools/p16_probe.py [n]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from saragan_amd import _lib  # noqa: E402
from saragan_amd._lib import ConvEpilogue, ConvShape  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d, h, w = 32, 128, 128
fl = 2.0 * n * d * h * w * 64 * 32 * 27


def setup(ups):
    shp = ConvShape(n, d, h, w, 64, 32, 3, 3, 3, 1 if ups else 0)
    x = torch.randn((n, d // 2, h // 2, w // 2, 64) if ups else (n, d, h, w, 64), device=dev).bfloat16()
    wt = torch.randn(3, 3, 3, 32, 64, device=dev)
    wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
    _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 1, wp.data_ptr(), C.byref(shp), dt, st))
    y = torch.empty(n, d, h, w, 32, device=dev, dtype=torch.bfloat16)
    ws = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
    return shp, x, wp, y, torch.empty(max(16, ws), device=dev, dtype=torch.uint8), ws


bias = torch.zeros(32, device=dev)
bits32 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 1), device=dev, dtype=torch.int32)
bits64 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 2), device=dev, dtype=torch.int32)
sout = torch.empty_like(bits32)
for ups, mode in ((False, 'signs'), (False, 'mask'), (True, 'gather+mask')):
    shp, x, wp, y, wsb, ws = setup(ups)
    if mode == 'signs':
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
    else:
        ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits32.data_ptr(), 0.2, None)
    ep.workspace, ep.workspace_bytes = wsb.data_ptr(), ws
    if ups:
        ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = bits64.data_ptr(), 0.2, 0.125
    call = lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
    res, outs = {}, {}
    for rep in range(3):
        for v in (0, 1):
            os.environ['SG_FWD3P_16'] = str(v)
            lib.sg_config_reload()
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[v] = y.clone()
    ne = int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum())
    a_, b_ = min(res[0]), min(res[1])
    print(f'n{n} 64->32 {mode}: 32x32x16 {a_:8.1f} us ({fl / a_ / 1e6:6.0f} TF/s)   16x16x32 {b_:8.1f} us ({fl / b_ / 1e6:6.0f} TF/s)   {(a_ / b_ - 1) * 100:+.2f} %   '
          f'elements differing {ne} of {y.numel()}   all: {[round(t) for t in res[0]]} / {[round(t) for t in res[1]]}', flush=True)
os.environ['SG_FWD3P_16'] = '0'
lib.sg_config_reload()
