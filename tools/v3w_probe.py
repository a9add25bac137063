"""Diagnostic: the sliding-halo kernel (conv_fwd3s, v_mfma_f32_32x32x16_bf16, SG_FWD3S_16=0) against conv_fwd3w (wave-private planes,
sliding accumulators, v_mfma_f32_16x16x32_bf16), A/B in one process at the bench shapes, back-to-back launches on random data.
usage: python tools/v3w_probe.py [n] [d h w]"""
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
d, h, w = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (32, 128, 128)
reps = int(os.environ.get('REPS', '20'))

for cout, mode in ((32, 'signs'), (32, 'mask'), (32, 'pn+signs'), (64, 'signs'), (64, 'pool+signs'), (32, 'plain')):
    fl = 2.0 * n * d * h * w * 32 * cout * 27
    shp = ConvShape(n, d, h, w, 32, cout, 3, 3, 3, 0)
    x = torch.randn(n, d, h, w, 32, device=dev).bfloat16()
    wt = torch.randn(3, 3, 3, 32, cout, device=dev)
    wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
    _lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
    pool = 'pool' in mode
    y = torch.empty((n, d // 2, h, w // 2, cout) if pool else (n, d, h, w, cout), device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(cout, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, cout // 32), device=dev, dtype=torch.int32)
    sout = torch.empty_like(bits)
    scale = torch.empty(n * d * h * w, device=dev)
    if mode == 'plain':
        ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, None, 0.0, None)
    elif mode == 'mask':
        ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None)
    elif mode == 'pn+signs':
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 1, 1e-8, scale.data_ptr(), None, 0.0, sout.data_ptr())
    else:
        ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
    ep.pool = 1 if pool else 0
    call = lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
    res, outs = {}, {}
    for rep in range(3):
        for v in (0, 1):
            os.environ['SG_FWD3S_16'] = str(v)
            lib.sg_config_reload()
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / reps * 1e3)
            outs[v] = y.clone()
    ne = int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum())
    md = float((outs[0].float() - outs[1].float()).abs().max() / outs[0].float().abs().max())
    a_, b_ = min(res[0]), min(res[1])
    print(f'n{n} {d}x{h}x{w} 32->{cout} {mode}: fwd3s {a_:8.1f} us ({fl / a_ / 1e6:6.0f} TF/s)   fwd3w {b_:8.1f} us ({fl / b_ / 1e6:6.0f} TF/s)   {(a_ / b_ - 1) * 100:+.2f} %   '
          f'elements differing {ne} of {y.numel()} (max rel {md:.2e})   all: {[round(t) for t in res[0]]} / {[round(t) for t in res[1]]}', flush=True)
os.environ['SG_FWD3S_16'] = '1'
lib.sg_config_reload()
