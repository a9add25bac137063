from saragan_amd.networks.pgan.discriminator import *  # noqa: F401,F403
from saragan_amd.networks.pgan.discriminator import discriminator  # noqa: F401
