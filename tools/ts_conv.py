import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib
from saragan_amd._lib import ConvEpilogue, ConvShape
lib = _lib.load()
lib.sg_debug_set_ts_buffer.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
d, h, w = (int(t) for t in os.environ.get('TS_DHW', '32,128,128').split(','))
n, cin, cout = int(os.environ.get('TS_N', '16')), int(os.environ.get('TS_CIN', '32')), int(os.environ.get('TS_COUT', '32'))
shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
wt = torch.randn(3, 3, 3, cin, cout, device=dev)
wp = torch.empty(lib.sg_conv3d_packed_bytes(C.byref(shp), dt), device=dev, dtype=torch.uint8)
_lib.check(lib.sg_conv3d_pack_weights(wt.data_ptr(), 0.05, 0, wp.data_ptr(), C.byref(shp), dt, st))
y = torch.empty(n, d, h, w, cout, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(cout, device=dev)
mode = os.environ.get('TS_EPI', 'act')      # act | signs | mask | pool
nw = (cout + 31) // 32
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, d, h, w, nw), device=dev, dtype=torch.int64).to(torch.int32)
sout = torch.empty_like(bits)
if mode == 'mask':
    ep = ConvEpilogue(None, 0, 0.2, 0, 1e-8, None, bits.data_ptr(), 0.2, None)
elif mode == 'signs':
    ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
elif mode == 'pool':
    ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, sout.data_ptr())
    ep.pool = 1
else:
    ep = ConvEpilogue(bias.data_ptr(), 1, 0.2, 0, 1e-8, None, None, 0.0, None)
for _ in range(3):
    _lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
ts = torch.zeros(256, dtype=torch.int64, device=dev)
lib.sg_debug_set_ts_buffer(ts.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.sg_conv3d_fwd(x.data_ptr(), wp.data_ptr(), y.data_ptr(), C.byref(shp), C.byref(ep), dt, st))
e1.record()
torch.cuda.synchronize()
print('kernel ms', e0.elapsed_time(e1))
lib.sg_debug_set_ts_buffer(None)
t = ts.cpu().numpy()
for g in range(2):
    v = t[g * 128:(g + 1) * 128]
    v = v[v > 0]
    print('group', g, 'n', len(v), 'span', int(v[-1] - v[0]) if len(v) else 0)
    print(' deltas:', [int(b - a) for a, b in zip(v[:40], v[1:41])])
