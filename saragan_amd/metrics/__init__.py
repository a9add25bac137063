"""Download-free validation metrics of SURFGAN_3D/metrics computed on the GPU: sliced
Wasserstein distance over a Laplacian pyramid (swd.py) and MSE / NRMSE / PSNR / SSIM (skim_metrics.py).  FID stays out:
it needs the Inception graph the reference downloads (metrics/fid_new.py:291-318)."""
