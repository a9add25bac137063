"""The 2-D twin's network layer (SURFGAN_2D/networks): `ops` with conv2d / upscale2d / downscale2d and the `pgan`
generator / discriminator with the 2-D tree's legacy signature (num_phases, base_dim, size).  BASELINE config 5 (2-D
1024^2, fp32) is a network / kernel-level target (the 2-D reference LOOP does not run at
head); images are NCHW, computed as D == 1 volumes on the same gfx950 kernels."""
