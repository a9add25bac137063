import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from saragan_amd import functional as F
from oracle import pgan_oracle as O
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(61)
n, cin, cout, sp = 2, 32, 64, (6, 128, 256)
x = torch.randn((n, cin, *sp), generator=g).bfloat16()
w = torch.randn((3, 3, 3, cin, cout), generator=g)
b = torch.randn(cout, generator=g) * 0.3
coef = O.runtime_coef(w.shape, 'leaky_relu', 0.2)
xg = x.to(dev).contiguous(memory_format=torch.channels_last_3d)
res = F.raw_conv(xg, w.to(dev), coef, False, bias=b.to(dev), act=True, slope=0.2, want_signs=True, pool=True)
ydw, _, signs = res
full, _, signs2 = F.raw_conv(xg, w.to(dev), coef, False, bias=b.to(dev), act=True, slope=0.2, want_signs=True)
ref = full.float()
ref = 0.25 * (ref[:, :, 0::2, :, 0::2] + ref[:, :, 1::2, :, 0::2] + ref[:, :, 0::2, :, 1::2] + ref[:, :, 1::2, :, 1::2])
err = (ydw.float() - ref).abs()
err = torch.nan_to_num(err, nan=1e30, posinf=1e30)
bad = (err > 0.05).nonzero().cpu().numpy()
print('ydw', tuple(ydw.shape), 'bad', len(bad), 'of', err.numel(), 'signs equal', bool((signs == signs2).all()))
for ax, nm in enumerate(['n', 'c', 'd', 'h', 'w']):
    vals, cnt = np.unique(bad[:, ax], return_counts=True)
    print(nm, dict(list(zip(vals.tolist(), cnt.tolist()))[:48]))
