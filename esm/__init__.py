"""Drop-in ``esm`` namespace: the names the reference package exports (reference esm/__init__.py),
served by the MI355X-native implementation in ``esm_amd``."""
from esm_amd import __version__  # noqa: F401
from esm_amd.alphabet import Alphabet, BatchConverter  # noqa: F401
from esm_amd.fasta import FastaBatchedDataset  # noqa: F401
from esm_amd.esm2 import ESM2  # noqa: F401
from esm_amd.msa_transformer import MSATransformer  # noqa: F401
from esm_amd.legacy import ProteinBertModel  # noqa: F401
from . import data, pretrained  # noqa: F401
