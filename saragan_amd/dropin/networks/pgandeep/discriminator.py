from saragan_amd.networks.pgandeep.discriminator import *  # noqa: F401,F403
from saragan_amd.networks.pgandeep.discriminator import discriminator  # noqa: F401
