from esm_amd.esm2 import ESM2  # noqa: F401
