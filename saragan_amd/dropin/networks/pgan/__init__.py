"""networks.pgan of the reference tree (networks/pgan/*.py)."""
