"""Diagnostic: the gathered two-pass 64 -> 32 data gradient (conv_fwd3s<2, 1|2, ..., UPS, INM>) with parts of its off-phase
switched off (SG_DBG_FLAGS: 1 no halo staging, 2 no epilogue, 256 no output mask) -- where the off-phase's time goes.
Results are garbage under the flags; only the durations mean something.  usage: python tools/ksplit_probe.py [n]"""
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
vox = n * d * h * w
gyh = torch.randn(n, d // 2, h // 2, w // 2, 64, device=dev).to(torch.bfloat16)
bits64 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 2), device=dev, dtype=torch.int32)
bits32 = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, 1), device=dev, dtype=torch.int32)
wt = torch.randn(3, 3, 3, 32, 64, device=dev)
shp = ConvShape(n, d, h, w, 64, 32, 3, 3, 3, 1)
wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
_lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 1, wp.data_ptr(), C.byref(shp), dt, st))
gx = torch.empty(n, d, h, w, 32, device=dev, dtype=torch.bfloat16)
fws_bytes = lib.sg_conv3d_fwd_workspace(C.byref(shp), dt)
fws = torch.empty(max(16, fws_bytes), device=dev, dtype=torch.uint8)


def run(masked_in, masked_out):
    ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, bits32.data_ptr() if masked_out else None, 0.2, None)
    ep.workspace, ep.workspace_bytes = fws.data_ptr(), fws_bytes
    if masked_in:
        ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = bits64.data_ptr(), 0.2, 0.125
    call = lambda: _lib.check(lib.sg_conv3d_fwd(gyh.data_ptr(), wp.data_ptr(), gx.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


flops = 2.0 * vox * 64 * 32 * 27
for flags in (0, 1, 2, 3, 256):
    os.environ['SG_DBG_FLAGS'] = str(flags)
    lib.sg_config_reload()
    a, b = run(True, True), run(False, True)
    print(f'SG_DBG_FLAGS={flags:3d}: gathered+in-mask {a:8.1f} us ({flops / a / 1e6:6.0f} TF/s)   gathered, no in-mask {b:8.1f} us ({flops / b / 1e6:6.0f} TF/s)', flush=True)

# in-kernel stamps of block 8, wave 0 of each group (second pass: it runs last and overwrites the first pass's stamps)
lib.sg_debug_set_ts_buffer.argtypes = [C.c_void_p]
os.environ['SG_DBG_FLAGS'] = '128'
lib.sg_config_reload()
ts = torch.zeros(256, dtype=torch.int64, device=dev)
ep = ConvEpilogue(None, 0, 0.0, 0, 1e-8, None, bits32.data_ptr(), 0.2, None)
ep.workspace, ep.workspace_bytes = fws.data_ptr(), fws_bytes
ep.in_mask_bits, ep.in_mask_slope, ep.in_gain = bits64.data_ptr(), 0.2, 0.125
lib.sg_debug_set_ts_buffer(ts.data_ptr())
_lib.check(lib.sg_conv3d_fwd(gyh.data_ptr(), wp.data_ptr(), gx.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
torch.cuda.synchronize()
lib.sg_debug_set_ts_buffer(None)
t = ts.cpu().numpy()
for g in range(2):
    v = t[g * 128:(g + 1) * 128]
    v = v[v > 0]
    print('group', g, 'stamps', len(v), 'deltas (100 MHz ticks x ? -- s_memtime):', [int(b - a) for a, b in zip(v[:48], v[1:49])])
