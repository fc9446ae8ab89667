"""MSA Transformer entry points (reference esm/model/msa_transformer.py).  The axial row/column
attention path (config 5) is not built yet in this round; the class exists so that
``isinstance(model, MSATransformer)`` checks in reference scripts (scripts/extract.py:66-69)
stay meaningful."""
import torch.nn as nn


class MSATransformer(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "MSATransformer (axial attention) is not implemented yet by the MI355X engine"
        )


def build_from_checkpoint(model_data):
    raise NotImplementedError("MSA Transformer checkpoints are not supported yet by the MI355X engine")
