"""networks/ops.py of the reference tree -> saragan_amd.networks.ops."""
from saragan_amd.networks.ops import *  # noqa: F401,F403
