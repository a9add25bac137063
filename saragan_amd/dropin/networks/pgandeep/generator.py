from saragan_amd.networks.pgandeep.generator import *  # noqa: F401,F403
from saragan_amd.networks.pgandeep.generator import generator  # noqa: F401
