"""Drop-in counterparts of the reference's attack_models/ scripts (fbb.py, utils.py, eval_roc.py)."""
