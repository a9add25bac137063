from saragan_amd.networks.pgan.generator import *  # noqa: F401,F403
from saragan_amd.networks.pgan.generator import generator  # noqa: F401
