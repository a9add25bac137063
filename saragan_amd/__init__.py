"""saragan_amd: MI355X-native (gfx950) engine for the SURFGAN_3D `pgan` generator + discriminator
training step of sara-nl/saraGAN.  HIP kernels + C ABI in csrc/ and include/saragan_hip.h; the Python
modules mirror the reference's interface (networks.ops, networks.pgan.*, networks.loss, optimization)."""
__version__ = '0.2.0'


def dropin_path():
    """Directory to put on sys.path so that the reference loop's module paths (`networks.pgan.generator`,
    `optimization`, `dataset`, ... as imported by SURFGAN_3D/optuna_objective.py:12-30,64-65) resolve to this package."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dropin')
