from esm_amd.alphabet import RESIDUE_SYMBOLS

proteinseq_toks = {"toks": list(RESIDUE_SYMBOLS)}
