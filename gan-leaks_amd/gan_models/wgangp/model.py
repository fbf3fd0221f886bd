"""Counterpart of the reference's gan_models/wgangp/model.py: its Generator (model.py:37-58) has the
same layers and state_dict keys as the DCGAN one (ReLU(inplace=True) is the only difference), so it
shares the implementation."""
from ..dcgan.model_torch import Generator  # noqa: F401
