"""In-kernel phase stamps of the sliding-halo weight-gradient kernel (conv_wgrad3): per wave group, cycles of
[MFMA phase | barrier | staging issue | DMA landed | barrier].  TS_N / TS_CIN / TS_COUT choose the layer."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from saragan_amd import _lib
from saragan_amd._lib import ConvShape
lib = _lib.load()
lib.sg_debug_set_ts_buffer.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
n, d, h, w = int(os.environ.get('TS_N', '32')), 32, 128, 128
cin, cout = int(os.environ.get('TS_CIN', '32')), int(os.environ.get('TS_COUT', '64'))
shp = ConvShape(n, d, h, w, cin, cout, 3, 3, 3, 0)
dt = _lib.SG_BF16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
dy = torch.randn(n, d, h, w, cout, device=dev).bfloat16()
dw = torch.empty(3, 3, 3, cin, cout, device=dev)
wsb = lib.sg_conv3d_wgrad_workspace(C.byref(shp), dt)
ws = torch.empty(wsb, device=dev, dtype=torch.uint8)


def call():
    _lib.check(lib.sg_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0.05, ws.data_ptr(), wsb, C.byref(shp), dt, st))


for _ in range(3):
    call()
ts = torch.zeros(256, dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
print('kernel ms (no stamps)', e0.elapsed_time(e1))
lib.sg_debug_set_ts_buffer(ts.data_ptr())
e0.record(); call(); e1.record(); torch.cuda.synchronize()
print('kernel ms (stamps, DMA waited inside the off-phase)', e0.elapsed_time(e1))
lib.sg_debug_set_ts_buffer(None)
t = ts.cpu().numpy()
for g in range(2):
    v = t[g * 128:(g + 1) * 128]
    v = v[v > 0]
    print('group', g, 'n', len(v), 'span', int(v[-1] - v[0]) if len(v) else 0)
    print(' deltas:', [int(b - a) for a, b in zip(v[:40], v[1:41])])
