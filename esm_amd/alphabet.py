"""Vocabulary and batch tokenisation with the behaviour of the reference's ``esm.data.Alphabet``,
``BatchConverter`` and ``MSABatchConverter`` (reference esm/data.py:91-336, esm/constants.py:7-9).

Written from the behavioural spec, not from the reference source: the
tokeniser is a single compiled regular expression over the vocabulary plus a byte lookup table
for the common case of a plain amino-acid string, and batches are filled through numpy.
"""
import re
from typing import List, Sequence, Tuple

import numpy as np
import torch

# the 27 residue / gap symbols of every ESM alphabet, in id order (esm/constants.py:7-9)
RESIDUE_SYMBOLS = list("LAGVSERTIDPKQNFYMHWCXBUZO.-")

_ARCHITECTURES = {
    # name -> (prepend, append, prepend_bos, append_eos, use_msa)      (esm/data.py:142-174)
    "ESM-1": (("<null_0>", "<pad>", "<eos>", "<unk>"), ("<cls>", "<mask>", "<sep>"), True, False, False),
    "ESM-1b": (("<cls>", "<pad>", "<eos>", "<unk>"), ("<mask>",), True, True, False),
    "MSA Transformer": (("<cls>", "<pad>", "<eos>", "<unk>"), ("<mask>",), True, False, True),
    "invariant_gvp": (("<null_0>", "<pad>", "<eos>", "<unk>"), ("<mask>", "<cath>", "<af2>"), True, False, False),
}
_ALIASES = {"protein_bert_base": "ESM-1", "roberta_large": "ESM-1b", "msa_transformer": "MSA Transformer"}


class Alphabet:
    """Token <-> id mapping.  ``all_toks`` = prepend + residues + ``<null_i>`` fill to a multiple
    of 8 + append (esm/data.py:108-112)."""

    def __init__(
        self,
        standard_toks: Sequence[str],
        prepend_toks: Sequence[str] = ("<null_0>", "<pad>", "<eos>", "<unk>"),
        append_toks: Sequence[str] = ("<cls>", "<mask>", "<sep>"),
        prepend_bos: bool = True,
        append_eos: bool = False,
        use_msa: bool = False,
    ):
        self.standard_toks = list(standard_toks)
        self.prepend_toks = list(prepend_toks)
        self.append_toks = list(append_toks)
        self.prepend_bos = prepend_bos
        self.append_eos = append_eos
        self.use_msa = use_msa

        toks = self.prepend_toks + self.standard_toks
        n_fill = -len(toks) % 8
        toks += [f"<null_{i + 1}>" for i in range(n_fill)]
        toks += self.append_toks
        self.all_toks = toks
        self.tok_to_idx = {t: i for i, t in enumerate(toks)}

        self.unk_idx = self.tok_to_idx["<unk>"]
        self.padding_idx = self.get_idx("<pad>")
        self.cls_idx = self.get_idx("<cls>")
        self.mask_idx = self.get_idx("<mask>")
        self.eos_idx = self.get_idx("<eos>")
        self.all_special_tokens = ["<eos>", "<unk>", "<pad>", "<cls>", "<mask>"]
        self.unique_no_split_tokens = self.all_toks

        # tokeniser: longest vocabulary entries first, then whitespace, then any other character
        alts = sorted(set(toks), key=len, reverse=True)
        self._scan = re.compile("|".join(re.escape(t) for t in alts) + r"|\s+|.", re.DOTALL)
        # byte -> id table for strings made only of one-character tokens
        lut = np.full(256, -1, dtype=np.int64)
        for t, i in self.tok_to_idx.items():
            if len(t) == 1 and ord(t) < 256:
                lut[ord(t)] = i
        self._lut = lut

    def __len__(self):
        return len(self.all_toks)

    def get_idx(self, tok):
        return self.tok_to_idx.get(tok, self.unk_idx)

    def get_tok(self, ind):
        return self.all_toks[ind]

    def to_dict(self):
        return dict(self.tok_to_idx)

    def get_batch_converter(self, truncation_seq_length: int = None):
        cls = MSABatchConverter if self.use_msa else BatchConverter
        return cls(self, truncation_seq_length)

    @classmethod
    def from_architecture(cls, name: str) -> "Alphabet":
        key = _ALIASES.get(name, name)
        if key not in _ARCHITECTURES and "invariant_gvp" in name.lower():
            key = "invariant_gvp"
        if key not in _ARCHITECTURES:
            raise ValueError("Unknown architecture selected")
        prepend, append, bos, eos, msa = _ARCHITECTURES[key]
        return cls(RESIDUE_SYMBOLS, prepend, append, bos, eos, msa)

    def _tokenize(self, text) -> List[str]:
        return text.split()

    def tokenize(self, text, **kwargs) -> List[str]:
        """Split ``text`` into vocabulary tokens; whitespace separates and is dropped; characters
        that are not part of any token are returned as maximal runs (``encode`` then raises
        ``KeyError`` for them, like the reference, esm/data.py:249-250)."""
        out: List[str] = []
        run: List[str] = []
        for m in self._scan.finditer(text):
            piece = m.group(0)
            if piece in self.tok_to_idx:
                if run:
                    out.append("".join(run))
                    run = []
                out.append(piece)
            elif piece.isspace():
                if run:
                    out.append("".join(run))
                    run = []
            else:
                run.append(piece)
        if run:
            out.append("".join(run))
        return out

    def encode(self, text) -> List[int]:
        return self.encode_array(text).tolist()

    def encode_array(self, text) -> np.ndarray:
        """``encode`` as an int64 numpy array; plain residue strings take the table path."""
        if text.isascii() and "<" not in text and not any(c.isspace() for c in text):
            ids = self._lut[np.frombuffer(text.encode("ascii"), dtype=np.uint8)]
            if ids.size == 0 or ids.min() >= 0:
                return ids
        # general path; unknown symbols raise KeyError (they are not mapped to <unk>)
        return np.array([self.tok_to_idx[t] for t in self.tokenize(text)], dtype=np.int64)


class BatchConverter:
    """[(label, sequence)] -> (labels, sequences, int64 tokens [B, Lmax + bos + eos])
    (reference esm/data.py:253-297): cls first, ids, eos right after the ids, pad elsewhere;
    truncation applies to the encoded ids, the returned strings stay untruncated."""

    def __init__(self, alphabet, truncation_seq_length: int = None):
        self.alphabet = alphabet
        self.truncation_seq_length = truncation_seq_length

    def __call__(self, raw_batch: Sequence[Tuple[str, str]]):
        a = self.alphabet
        labels = [lab for lab, _ in raw_batch]
        strs = [s for _, s in raw_batch]
        encoded = [a.encode_array(s) for s in strs]
        if self.truncation_seq_length:
            encoded = [e[: self.truncation_seq_length] for e in encoded]
        bos, eos = int(a.prepend_bos), int(a.append_eos)
        width = max(len(e) for e in encoded) + bos + eos
        toks = np.full((len(encoded), width), a.padding_idx, dtype=np.int64)
        for row, e in zip(toks, encoded):
            if bos:
                row[0] = a.cls_idx
            row[bos : bos + len(e)] = e
            if eos:
                row[bos + len(e)] = a.eos_idx
        return labels, strs, torch.from_numpy(toks)


class MSABatchConverter(BatchConverter):
    """One MSA ([(label, row)]) or a list of MSAs -> tokens [B, depth_max, width_max + bos + eos]
    (reference esm/data.py:300-336); rows of one MSA must have equal length."""

    def __call__(self, inputs):
        batch = [inputs] if isinstance(inputs[0][0], str) else inputs
        a = self.alphabet
        depth = max(len(msa) for msa in batch)
        width = max(len(msa[0][1]) for msa in batch) + int(a.prepend_bos) + int(a.append_eos)
        tokens = torch.full((len(batch), depth, width), a.padding_idx, dtype=torch.int64)
        labels, strs = [], []
        for i, msa in enumerate(batch):
            if len({len(seq) for _, seq in msa}) != 1:
                raise RuntimeError(
                    "Received unaligned sequences for input to MSA, all sequence lengths must be equal."
                )
            lab, st, tk = BatchConverter.__call__(self, msa)
            labels.append(lab)
            strs.append(st)
            tokens[i, : tk.size(0), : tk.size(1)] = tk
        return labels, strs, tokens
