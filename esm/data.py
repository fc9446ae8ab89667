"""``esm.data`` names (reference esm/data.py) -> esm_amd implementations."""
from esm_amd.alphabet import Alphabet, BatchConverter, MSABatchConverter  # noqa: F401
from esm_amd.fasta import FastaBatchedDataset, read_alignment_lines, read_fasta  # noqa: F401
