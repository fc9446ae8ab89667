"""Host side of token-packed batches (include/esmk.h: esmk_forward_packed).

The reference pads every batch to its longest member (esm/data.py:269-277); the engine can instead take the
sequences back to back in one row space.  Layout rule (checked again by the C ABI): segment starts are
multiples of 16 rows, the row count is a multiple of 128, rows between segments hold the padding index.

Only the [B,2] segment table is computed on the host; tokens are packed and results unpacked on the device with
index tensors derived from it there, so a forward_varlen call queues without waiting for the previous one."""
from dataclasses import dataclass

import torch

SEG_ALIGN = 16    # V^T keeps keys permuted inside groups of 16 (esm_amd/csrc/attention.hip)
ROWS_ALIGN = 128  # whole 128-row wave blocks in the GEMM epilogues


def _to_device(t, device):
    device = torch.device(device)
    if device.type == "cuda" and not t.is_cuda:
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


@dataclass
class PackPlan:
    B: int
    T: int
    rows: int
    lengths: torch.Tensor   # int64 [B] (CPU)
    segments: torch.Tensor  # int32 [B,2] (CPU, contiguous): first row, length

    def index(self, device):
        """(idx, keep) on ``device``: idx[b,t] = packed row of token (b,t), or ``rows`` (one scratch slot behind
        the row space) where keep[b,t] is False."""
        seg = _to_device(self.segments, device).to(torch.int64)
        t = torch.arange(self.T, device=device).unsqueeze(0)
        keep = t < seg[:, 1:2]
        idx = torch.where(keep, seg[:, 0:1] + t, torch.full_like(t, self.rows))
        return idx, keep

    def pack(self, tokens, padding_idx, idx):
        """[B,T] tokens -> int64 [rows] on the device of ``idx``."""
        flat = torch.full((self.rows + 1,), padding_idx, dtype=torch.int64, device=idx.device)
        flat.scatter_(0, idx.reshape(-1), _to_device(tokens, idx.device).to(torch.int64).reshape(-1))
        return flat[: self.rows]

    def unpack(self, x, idx, keep):
        """packed [rows, C] -> padded [B, T, C] with zeros where keep is False."""
        out = x.index_select(0, idx.clamp(max=self.rows - 1).reshape(-1)).view(self.B, self.T, x.shape[1])
        return out.masked_fill_(~keep.unsqueeze(-1), 0)


def pack_plan(tokens, padding_idx, lengths=None):
    """Segment layout for a right-padded [B,T] batch.  ``lengths`` defaults to 1 + index of the last non-pad
    token of each row (interior <pad> tokens stay inside their segment and are masked by the engine); reading
    them from a device tensor costs a device synchronisation, from a CPU tensor nothing."""
    B, T = tokens.shape
    if lengths is None:
        ar = torch.arange(1, T + 1, device=tokens.device)
        lengths = (tokens.ne(padding_idx) * ar).amax(dim=1)
    lengths = torch.as_tensor(lengths).to("cpu", torch.int64).clamp(min=1, max=T)
    padded = (lengths + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN
    starts = torch.cumsum(padded, 0) - padded
    rows = int((int(padded.sum()) + ROWS_ALIGN - 1) // ROWS_ALIGN * ROWS_ALIGN)
    segments = torch.stack([starts, lengths], dim=1).to(torch.int32).contiguous()
    return PackPlan(B, T, rows, lengths, segments)
