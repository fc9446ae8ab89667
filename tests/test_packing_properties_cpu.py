"""Property tests (hypothesis) of the host side of token-packed batches (esm_amd/packing.py) and of the batch planner of
the extraction driver (esm_amd/extract.py): invariants the engine relies on (esmk_forward_packed's segment-table
contract, include/esmk.h) for arbitrary length mixes."""
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from esm_amd.extract import assign_batches
from esm_amd.fasta import FastaBatchedDataset
from esm_amd.packing import pack_plan


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(min_value=1, max_value=300), min_size=1, max_size=12), st.integers(0, 2 ** 31 - 1))
def test_pack_plan_invariants(lengths, seed):
    T = max(lengths)
    g = torch.Generator().manual_seed(seed)
    toks = torch.randint(4, 24, (len(lengths), T), generator=g)
    for b, n in enumerate(lengths):
        toks[b, n:] = 1
        toks[b, n - 1] = 2  # every row ends in a non-pad token so that the inferred length is n
    plan = pack_plan(toks, 1)
    seg = plan.segments
    assert seg.dtype == torch.int32 and seg.shape == (len(lengths), 2)
    assert plan.lengths.tolist() == lengths and seg[:, 1].tolist() == lengths
    starts = seg[:, 0].tolist()
    assert starts[0] == 0 and all(s % 16 == 0 for s in starts)
    assert all(starts[i] + lengths[i] <= starts[i + 1] for i in range(len(lengths) - 1))  # ascending, disjoint
    assert plan.rows % 128 == 0 and starts[-1] + lengths[-1] <= plan.rows
    assert plan.rows - (starts[-1] + lengths[-1]) < 128 + 16  # no more padding than the alignment asks for
    idx, keep = plan.index("cpu")
    flat = plan.pack(toks, 1, idx)
    assert flat.shape == (plan.rows,)
    for b, n in enumerate(lengths):
        assert torch.equal(flat[starts[b]:starts[b] + n], toks[b, :n])
    assert int(flat.ne(1).sum()) == int(toks.ne(1).sum())  # gaps hold <pad>
    back = plan.unpack(flat.unsqueeze(1).float(), idx, keep).squeeze(-1).long()
    assert torch.equal(back[keep], toks[keep]) and int(back[~keep].abs().sum()) == 0


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(min_value=1, max_value=1200), min_size=1, max_size=40), st.integers(1, 8),
       st.sampled_from([512, 1024, 4096, 65536]))
def test_batch_plan_covers_every_sequence_once(lengths, world, toks_per_batch):
    ds = FastaBatchedDataset([f"s{i}" for i in range(len(lengths))], ["A" * n for n in lengths])
    batches = ds.get_batch_indices(toks_per_batch, extra_toks_per_seq=1)
    assert sorted(i for b in batches for i in b) == list(range(len(lengths)))
    for b in batches:  # token budget: only a single over-long sequence may exceed it (reference esm/data.py:65-88)
        width = max(lengths[i] for i in b) + 1
        assert len(b) == 1 or len(b) * width <= toks_per_batch
    plan = assign_batches(batches, lengths, world, 1280)
    assert sorted(x for r in plan for x in r) == list(range(len(batches)))
    assert plan == assign_batches(batches, lengths, world, 1280)  # every rank computes the same plan
