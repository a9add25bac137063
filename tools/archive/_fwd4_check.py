import os, sys, torch
sys.path.insert(0, '/root/repo')
import ctypes as C
from saragan_amd import _lib, functional as F
lib = _lib.load()
torch.manual_seed(0)
def kernels():
    ents = (_lib.ProfEntry * 16)(); cnt = C.c_int32(0)
    lib.sg_prof_collect(ents, 16, C.byref(cnt))
    return sorted({ents[i].kernel.decode() for i in range(cnt.value)})
for (n, cin, cout) in ((64, 128, 512), (64, 512, 128), (64, 128, 128), (32, 128, 512)):
    x = torch.randn(n, cin, 4, 16, 16, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(3, 3, 3, cin, cout, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda') * 0.1
    coef = 0.02
    for mode in ('bias+lrelu+signs', 'flip+mask'):
        lib.sg_prof_enable(1)
        with torch.no_grad():
            if mode == 'bias+lrelu+signs':
                y, _, signs = F.raw_conv(x, w, coef, False, False, bias=b, act=True, slope=0.2, want_signs=True)
                wq = (w * coef).bfloat16().float()
                ref = torch.nn.functional.conv3d(x.float(), wq.permute(4, 3, 0, 1, 2), padding=1) + b.view(1, -1, 1, 1, 1)
                ref = torch.nn.functional.leaky_relu(ref, 0.2)
                sref = F.sign_words(ref.bfloat16().contiguous(memory_format=torch.channels_last_3d))
            else:
                gy = torch.randn(n, cout, 4, 16, 16, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last_3d)
                bits = F.sign_words(x)
                y, _, _ = F.raw_conv(gy, w, coef, True, False, mask_bits=bits, mask_slope=0.2)
                wq = (w * coef).bfloat16().float()
                ref = torch.nn.functional.conv_transpose3d(gy.float(), wq.permute(4, 3, 0, 1, 2), padding=1)
                ref = torch.where(x.float() < 0, ref * 0.2, ref)
        torch.cuda.synchronize()
        k = kernels(); lib.sg_prof_enable(0)
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        fin = bool(torch.isfinite(y.float()).all())
        extra = ''
        if mode == 'bias+lrelu+signs':
            extra = f' sign mismatch {float((signs != sref).float().mean()):.2e}'
        print(n, cin, cout, mode, k, f'rel max err {err:.3e} finite {fin}{extra}', flush=True)
