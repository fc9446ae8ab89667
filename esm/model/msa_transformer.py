from esm_amd.msa_transformer import MSATransformer  # noqa: F401
