"""Independent numpy loop restatement of tf.nn.conv3d(x, w, strides=1, padding='SAME',
data_format='NCDHW') with a DHWIO filter, as called at SURFGAN_3D/networks/ops.py:147-150.
TEST INFRASTRUCTURE ONLY (pins oracle/pgan_oracle.py's F.conv3d mapping); small cases only.

TF 'SAME' with stride 1: out size = in size, pad_total = k-1, pad_before = (k-1)//2.
y[n,co,d,h,w] = sum_{i,j,l,ci} x[n,ci,d+i-pb_d,h+j-pb_h,w+l-pb_w] * f[i,j,l,ci,co]  (cross-correlation).
"""
import numpy as np


def conv3d_same_dhwio(x: np.ndarray, f: np.ndarray) -> np.ndarray:
    n, ci, D, H, W = x.shape
    kd, kh, kw, fci, co = f.shape
    assert fci == ci
    pd, ph, pw = (kd - 1) // 2, (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((n, ci, D + kd - 1, H + kh - 1, W + kw - 1), dtype=np.float64)
    xp[:, :, pd:pd + D, ph:ph + H, pw:pw + W] = x
    y = np.zeros((n, co, D, H, W), dtype=np.float64)
    for i in range(kd):
        for j in range(kh):
            for l in range(kw):
                patch = xp[:, :, i:i + D, j:j + H, l:l + W]           # [n,ci,D,H,W]
                y += np.einsum('ncdhw,co->nodhw', patch, f[i, j, l].astype(np.float64))
    return y
