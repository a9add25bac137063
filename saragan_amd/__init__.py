"""saragan_amd: MI355X-native (gfx950) engine for the SURFGAN_3D `pgan` generator + discriminator
training step of sara-nl/saraGAN.  HIP kernels + C ABI in csrc/ and include/saragan_hip.h; the Python
modules mirror the reference's interface (networks.ops, networks.pgan.*, networks.loss, optimization)."""
__version__ = '0.1.0'
