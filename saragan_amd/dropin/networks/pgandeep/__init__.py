"""networks.pgandeep of the reference tree (networks/pgandeep/*.py)."""
