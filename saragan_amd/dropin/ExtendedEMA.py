"""ExtendedEMA.py of the reference tree -> saragan_amd.ExtendedEMA."""
from saragan_amd.ExtendedEMA import *  # noqa: F401,F403
