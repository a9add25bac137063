"""saragan_amd: MI355X-native (gfx950) engine for the SURFGAN_3D `pgan` generator + discriminator
training step of sara-nl/saraGAN.  HIP kernels + C ABI in csrc/ and include/saragan_hip.h; the Python
modules mirror the reference's interface (networks.ops, networks.pgan.*, networks.loss, optimization)."""
__version__ = '0.2.0'


def dropin_path():
    """Directory to put on sys.path so that the reference loop's module paths (`networks.pgan.generator`,
    `optimization`, `dataset`, ... as imported by SURFGAN_3D/optuna_objective.py:12-30,64-65) resolve to this package."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dropin')


def set_deterministic(on=True):
    """Reproducible mode of the library (SG_DETERMINISTIC): weight-gradient kernels store per-block slabs that are added in
    a fixed order instead of using float atomics, and the gradient-penalty row sums are ordered -- two runs from the same
    state produce bit-identical weights (tests/test_deterministic_gpu.py).  Costs workspace (one slab per block of the
    weight-gradient grids) and a few percent of the step; off by default."""
    import os
    from . import _lib
    os.environ['SG_DETERMINISTIC'] = '1' if on else '0'
    _lib.load().sg_config_reload()
    # the kept filter-gradient workspaces are clean only under the mode that used them last (the reproducible finalize pass does
    # not clear the slabs it read): none survives a switch
    from . import functional as F
    F.clear_kept_workspaces()
