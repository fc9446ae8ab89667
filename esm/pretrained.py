"""``esm.pretrained`` names (reference esm/pretrained.py) -> esm_amd.checkpoint."""
from esm_amd.checkpoint import *  # noqa: F401,F403
from esm_amd.checkpoint import (  # noqa: F401
    _download_model_and_regression_data,
    _has_regression_weights,
    load_hub_workaround,
    load_model_and_alphabet,
    load_model_and_alphabet_core,
    load_model_and_alphabet_hub,
    load_model_and_alphabet_local,
    load_regression_hub,
)
import esm_amd.checkpoint as _c

globals().update({k: getattr(_c, k) for k in _c._RELEASED})
