"""FASTA input for bulk extraction with the behaviour of the reference's
``FastaBatchedDataset`` / ``read_fasta`` / ``read_alignment_lines`` (reference esm/data.py:19-88,
339-378), written from the behavioural spec in SURVEY.md Appendix A.
"""
import re
from typing import Iterable, Iterator, List, Tuple


def _iter_records(lines: Iterable[str]):
    """Yield (header_line_index, header_text_without_'>', [sequence lines]) per record.
    Text in front of the first header is kept and ends up in the first record, as in the
    reference (its line buffer is only cleared when a record is emitted, esm/data.py:30-37)."""
    head_idx, head, body = None, None, []
    for idx, line in enumerate(lines):
        if line.startswith(">"):
            if head is not None:
                yield head_idx, head, body
                body = []
            head_idx, head = idx, line[1:]
        else:
            body.append(line)
    if head is not None:
        yield head_idx, head, body


class FastaBatchedDataset:
    """Labels + sequences of a FASTA file and token-budget batching."""

    def __init__(self, sequence_labels, sequence_strs):
        self.sequence_labels = list(sequence_labels)
        self.sequence_strs = list(sequence_strs)

    @classmethod
    def from_file(cls, fasta_file):
        labels: List[str] = []
        seqs: List[str] = []
        with open(fasta_file, "r") as fh:
            for idx, head, body in _iter_records(fh):
                name = head.strip()
                # an empty header gets a synthetic label from its 0-based line number
                labels.append(name if name else f"seqnum{idx:09d}")
                seqs.append("".join(part.strip() for part in body))
        assert len(set(labels)) == len(labels), "Found duplicate sequence labels"
        return cls(labels, seqs)

    def __len__(self):
        return len(self.sequence_labels)

    def __getitem__(self, idx):
        return self.sequence_labels[idx], self.sequence_strs[idx]

    def get_batch_indices(self, toks_per_batch, extra_toks_per_seq=0):
        """Greedy packing of length-sorted sequences: a batch is closed as soon as adding the
        next sequence would make (longest member + extra) x (members + 1) exceed the budget; a
        sequence longer than the budget gets a batch of its own (reference esm/data.py:65-88)."""
        order = sorted((len(s), i) for i, s in enumerate(self.sequence_strs))
        batches: List[List[int]] = []
        cur: List[int] = []
        widest = 0
        for length, i in order:
            need = length + extra_toks_per_seq
            if cur and max(need, widest) * (len(cur) + 1) > toks_per_batch:
                batches.append(cur)
                cur, widest = [], 0
            widest = max(widest, need)
            cur.append(i)
        if cur:
            batches.append(cur)
        return batches


def read_alignment_lines(lines, keep_gaps=True, keep_insertions=True, to_upper=False) -> Iterator[Tuple[str, str]]:
    """Generator of (description, sequence) over FASTA / A3M text lines."""

    def clean(s):
        if not keep_gaps:
            s = s.replace("-", "")
        if not keep_insertions:
            s = re.sub("[a-z]", "", s)
        return s.upper() if to_upper else s

    desc = seq = None
    for line in lines:
        if line[:1] == ">":
            if seq is not None:
                yield desc, clean(seq)
            desc, seq = line.strip().lstrip(">"), ""
        else:
            assert isinstance(seq, str)
            seq += line.strip()
    assert isinstance(seq, str) and isinstance(desc, str)
    yield desc, clean(seq)


def read_fasta(path, keep_gaps=True, keep_insertions=True, to_upper=False):
    with open(path, "r") as fh:
        yield from read_alignment_lines(fh, keep_gaps=keep_gaps, keep_insertions=keep_insertions, to_upper=to_upper)
