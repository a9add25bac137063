"""In-kernel stamps of conv_fwd3w (csrc/conv3w.hip): per wave of block (8, 0): K loop / staging / epilogue cycles per plane and the
in-kernel clock (s_memtime against s_memrealtime at 100 MHz), after TS_WARM seconds of back-to-back launches on random data.
usage: TS_N=32 TS_COUT=32 TS_EPI=signs python tools/ts_conv3w.py"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from saragan_amd import _lib
from saragan_amd._lib import ConvEpilogue, ConvShape
lib = _lib.load()
lib.sg_debug_set_ts_buffer.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
d, h, w = (int(t) for t in os.environ.get('TS_DHW', '32,128,128').split(','))
n, cin, cout = int(os.environ.get('TS_N', '32')), 32, int(os.environ.get('TS_COUT', '32'))
shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
wt = torch.randn(3, 3, 3, cin, cout, device=dev)
wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
_lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(cout, device=dev)
mode = os.environ.get('TS_EPI', 'signs')      # plain | signs | mask
nw = (cout + 31) // 32
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int64).to(torch.int32)
sout = torch.empty_like(bits)
if mode == 'mask':
    ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None)
elif mode == 'signs':
    ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
else:
    ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, None, 0.0, None)
call = lambda: _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
t0 = time.time()
while time.time() - t0 < float(os.environ.get('TS_WARM', '2')):
    for _ in range(50):
        call()
    torch.cuda.synchronize()
ts = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
lib.sg_debug_set_ts_buffer(ts.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
call()
e1.record()
torch.cuda.synchronize()
print('kernel us', e0.elapsed_time(e1) * 1e3, ' TF/s', 2.0 * n * d * h * w * 32 * cout * 27 / e0.elapsed_time(e1) / 1e9)
lib.sg_debug_set_ts_buffer(None)
t = ts.cpu().numpy().reshape(8, 2, 128)
for wv in range(8):
    v, r = t[wv, 0], t[wv, 1]
    k = int((v > 0).sum())
    v, r = v[:k], r[:k]
    if k < 8:
        print('wave', wv, 'stamps', k)
        continue
    clk = (v[-1] - v[0]) / ((r[-1] - r[0]) / 100.0)      # cycles per microsecond = MHz
    dv = np.diff(v)
    # stamps: [loop start], then per phase: K end, staged, off end
    kl, stg, epi = dv[0::3], dv[1::3], dv[2::3]
    print(f'wave {wv}: {k} stamps, clock {clk:7.1f} MHz, span {int(v[-1] - v[0])} cyc;  K loop med {np.median(kl):7.0f} (min {kl.min()}, max {kl.max()})  '
          f'staging med {np.median(stg):6.0f} (max {stg.max()})  epilogue med {np.median(epi):6.0f} (max {epi.max()})')
    if wv == 0:
        print('   K loop cycles:', kl[:36].tolist())
        print('   staging      :', stg[:36].tolist())
        print('   epilogue     :', epi[:36].tolist())
