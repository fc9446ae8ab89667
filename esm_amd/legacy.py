"""``ProteinBertModel`` of the reference (esm/model/esm1.py): the ESM-1b / ESM-1v architecture runs on the
MI355X engine (esm_amd.esm1); the original ESM-1 architecture raises ``NotImplementedError``."""
from .esm1 import ProteinBertModel  # noqa: F401
