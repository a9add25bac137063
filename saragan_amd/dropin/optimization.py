"""optimization.py of the reference tree -> saragan_amd.optimization."""
from saragan_amd.optimization import *  # noqa: F401,F403
