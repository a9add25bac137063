"""`networks` as the reference's loop imports it (optuna_objective.py:64-65)."""
