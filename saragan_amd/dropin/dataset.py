"""dataset.py of the reference tree -> saragan_amd.dataset."""
from saragan_amd.dataset import *  # noqa: F401,F403
