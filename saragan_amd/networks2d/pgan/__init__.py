"""SURFGAN_2D/networks/pgan: the 2-D progressive GAN (legacy signature)."""
