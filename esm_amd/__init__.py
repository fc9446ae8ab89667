"""esm_amd — an MI355X (gfx950) native forward engine for ESM-2 behind the public Python surface
of facebookresearch/esm.  ``import esm`` (the shim package next to this one) re-exports the same
names, so reference scripts such as ``scripts/extract.py`` run unchanged.

Importing this package does not load the HIP library; ``esm_amd._native`` does, on the first
forward (or explicitly), and raises if libesmk.so is missing — there is no CPU fallback.
"""
from .alphabet import Alphabet, BatchConverter, MSABatchConverter  # noqa: F401
from .fasta import FastaBatchedDataset, read_alignment_lines, read_fasta  # noqa: F401
from .esm2 import ESM2  # noqa: F401
from . import checkpoint as pretrained  # noqa: F401

__version__ = "0.1.0"
