from esm_amd.legacy import ProteinBertModel  # noqa: F401
