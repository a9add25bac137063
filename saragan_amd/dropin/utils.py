"""utils.py of the reference tree -> saragan_amd.utils."""
from saragan_amd.utils import *  # noqa: F401,F403
