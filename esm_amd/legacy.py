"""Names of reference model families that are outside the MI355X engine's scope but must stay
importable because reference scripts import them (``from esm import ProteinBertModel``,
reference scripts/extract.py:12)."""
import torch.nn as nn


class ProteinBertModel(nn.Module):
    """ESM-1 / ESM-1b / ESM-1v (reference esm/model/esm1.py) — not implemented by this engine."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "ESM-1 family models are outside the scope of the MI355X ESM-2 engine (SURVEY.md §8 f-1)"
        )
