"""networks/loss.py of the reference tree -> saragan_amd.networks.loss."""
from saragan_amd.networks.loss import *  # noqa: F401,F403
