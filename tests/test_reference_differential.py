"""Live differential tests against the reference implementation itself — only where /root/reference is mounted
(the build container; skipped on the GPU box, where the committed fixtures of tests/golden/ take over).

The reference package is also called ``esm``, so it runs in a SUBPROCESS (tests/_reference_probe.py) that dumps
what the reference computes for seeded random inputs; this process then compares
  * the host side (esm_amd.Alphabet / BatchConverter / MSABatchConverter / FastaBatchedDataset.get_batch_indices /
    read_fasta) — integer work, bit-exact — with reference esm/data.py:19-378;
  * the oracles (oracle/esm2_oracle.py, esm1b_oracle.py, msa_oracle.py) with reference ESM2 / ProteinBertModel /
    MSATransformer forwards on synthetic weights at dims and seeds that are NOT among the committed fixtures
    (head dims 24 / 32 / 128, other depths), to <= 2e-5: the pin of the oracle is re-checked wherever the reference
    is available, not only on the frozen vectors.
"""
import os
import pickle
import subprocess
import sys

import pytest
import torch

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "esm")),
                                reason="/root/reference is not mounted here")


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    out = tmp_path_factory.mktemp("refprobe") / "probe.pkl"
    env = dict(os.environ, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_reference_probe.py"), str(out)],
                       capture_output=True, text=True, env=env, cwd=str(out.parent), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    with open(out, "rb") as f:
        return pickle.load(f)


def test_tokenizer_and_batch_converter_bit_exact(probe):
    import esm

    for arch, cases in probe["alphabet"].items():
        a = esm.Alphabet.from_architecture(arch)
        assert a.all_toks == cases["all_toks"], arch
        assert (a.padding_idx, a.cls_idx, a.eos_idx, a.unk_idx, a.mask_idx) == cases["ids"], arch
        assert (a.prepend_bos, a.append_eos) == cases["bos_eos"], arch
        for s, want in cases["encode"]:
            assert a.encode(s) == want, (arch, s)
        for s, want in cases["tokenize"]:
            assert a.tokenize(s) == want, (arch, s)
        for trunc, batch, (labels, strs, toks) in cases["batches"]:
            got = a.get_batch_converter(trunc)(batch)
            assert got[0] == labels and got[1] == strs, (arch, trunc)
            assert got[2].dtype == torch.int64 and torch.equal(got[2], toks), (arch, trunc)
        for s in cases["raises"]:
            with pytest.raises(KeyError):
                a.encode(s)


def test_msa_batch_converter_bit_exact(probe):
    import esm

    a = esm.Alphabet.from_architecture("msa_transformer")
    conv = a.get_batch_converter()
    for inp, (labels, strs, toks) in probe["msa_batches"]:
        got = conv(inp)
        assert got[0] == labels and got[1] == strs and torch.equal(got[2], toks)
    with pytest.raises(RuntimeError, match="unaligned"):
        conv([("a", "MKT"), ("b", "MK")])


def test_fasta_dataset_and_batching_bit_exact(probe, tmp_path):
    import esm

    for text, toks_per_batch, extra, (labels, seqs, batches) in probe["fasta"]:
        f = tmp_path / "x.fasta"
        f.write_text(text)
        ds = esm.FastaBatchedDataset.from_file(f)
        assert list(ds.sequence_labels) == labels and list(ds.sequence_strs) == seqs
        assert ds.get_batch_indices(toks_per_batch, extra_toks_per_seq=extra) == batches
    for path, kw, rows in probe["read_fasta"]:
        assert list(esm.data.read_fasta(path, **kw))[:len(rows)] == rows, kw


def _close(a, b, tol, mask=None):
    d = (a - b).abs()
    if mask is not None:
        d = d[mask]
    return d.numel() == 0 or d.max().item() <= tol


def test_esm2_oracle_matches_live_reference(probe):
    from esm_amd.synth import synth_esm2_state_dict
    from oracle.esm2_oracle import esm2_forward

    for c in probe["esm2"]:
        sd = synth_esm2_state_dict(c["L"], c["E"], c["H"], seed=c["seed"])
        out = esm2_forward(sd, c["tokens"], c["L"], c["H"], repr_layers=range(c["L"] + 1), return_contacts=True)
        nonpad = c["tokens"].ne(1)
        tag = (c["L"], c["E"], c["H"])
        assert _close(out["logits"], c["logits"], 2e-5, nonpad), tag
        for l, ref in c["representations"].items():
            assert _close(out["representations"][l], ref, 2e-5, nonpad), (tag, l)
        assert _close(out["attentions"], c["attentions"], 2e-6), tag
        assert _close(out["contacts"], c["contacts"], 2e-5), tag


def test_esm1b_oracle_matches_live_reference(probe):
    from esm_amd.synth import synth_esm1b_state_dict
    from oracle.esm1b_oracle import esm1b_forward

    for c in probe["esm1b"]:
        sd = synth_esm1b_state_dict(c["L"], c["E"], c["H"], seed=c["seed"], ln_before=c["ln_before"])
        out = esm1b_forward(sd, c["tokens"], c["L"], c["H"], repr_layers=range(c["L"] + 1))
        nonpad = c["tokens"].ne(1)
        assert _close(out["logits"], c["logits"], 2e-5, nonpad)
        for l, ref in c["representations"].items():
            assert _close(out["representations"][l], ref, 2e-5, nonpad), l


def test_msa_oracle_matches_live_reference(probe):
    from esm_amd.synth import synth_msa_state_dict
    from oracle.msa_oracle import msa_forward

    for c in probe["msa"]:
        sd = synth_msa_state_dict(c["L"], c["E"], c["H"], c["F"], seed=c["seed"])
        out = msa_forward(sd, c["tokens"], c["L"], c["H"], repr_layers=range(c["L"] + 1), return_contacts=True)
        nonpad = c["tokens"].ne(1)
        assert _close(out["logits"], c["logits"], 5e-5, nonpad)
        for l, ref in c["representations"].items():
            assert _close(out["representations"][l], ref, 5e-5, nonpad), l
        assert _close(out["row_attentions"], c["row_attentions"], 5e-6)
        assert _close(out["contacts"], c["contacts"], 5e-5)


def test_reference_extract_script_runs_unmodified_against_the_shim(tmp_path):
    """BASELINE configs[0]: the reference's scripts/extract.py (8M dims, examples FASTA, --nogpu), once inside the
    reference package and once — the SAME unmodified file — against this repo's ``esm`` package.  The shim run has the
    engine replaced by the oracle inside the test launcher only (tests/_run_reference_script.py): the product has no
    CPU path.  Same files, same labels, same shapes, values to oracle precision."""
    from esm_amd.synth import write_esm2_checkpoint

    ckpt = write_esm2_checkpoint(str(tmp_path), "esm2_t6_8M_UR50D", 6, 320, 20, seed=7)
    fasta = os.path.join(REFERENCE, "examples", "data", "few_proteins.fasta")
    script = os.path.join(REFERENCE, "scripts", "extract.py")
    args = [ckpt, fasta, "OUT", "--repr_layers", "0", "5", "6", "--include", "mean", "per_tok", "bos", "contacts",
            "--nogpu", "--toks_per_batch", "2048"]
    env = dict(os.environ, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    outs = {}
    for tag, cmd, pp in (("ref", [sys.executable, script], REFERENCE),
                         ("shim", [sys.executable, os.path.join(ROOT, "tests", "_run_reference_script.py"), script], ROOT)):
        out_dir = tmp_path / tag
        a = [x if x != "OUT" else str(out_dir) for x in args]
        r = subprocess.run(cmd + a, capture_output=True, text=True, env=dict(env, PYTHONPATH=pp), cwd=str(tmp_path),
                           timeout=900)
        assert r.returncode == 0, (tag, r.stderr[-3000:])
        outs[tag] = {p.name: torch.load(p, weights_only=False) for p in sorted(out_dir.glob("*.pt"))}
    assert outs["ref"] and sorted(outs["ref"]) == sorted(outs["shim"])
    for name, ref in outs["ref"].items():
        got = outs["shim"][name]
        assert got["label"] == ref["label"] and sorted(got) == sorted(ref)
        for key in ("representations", "mean_representations", "bos_representations"):
            assert sorted(got[key]) == sorted(ref[key]) == [0, 5, 6]
            for l in ref[key]:
                assert got[key][l].shape == ref[key][l].shape and got[key][l].dtype == ref[key][l].dtype
                assert (got[key][l] - ref[key][l]).abs().max().item() < 5e-5, (name, key, l)
        assert got["contacts"].shape == ref["contacts"].shape
        assert (got["contacts"] - ref["contacts"]).abs().max().item() < 5e-5


def test_reference_own_alphabet_tests_pass_on_our_alphabets():
    """The reference's OWN test helpers (tests/test_alphabet.py:6-45, loaded by file path — they import nothing at
    module level) executed on this repo's Alphabet objects; their public wrappers download checkpoints just to get an
    alphabet, which ``Alphabet.from_architecture`` provides offline (esm/pretrained.py:87-127 picks the same names)."""
    import importlib.util

    import esm

    spec = importlib.util.spec_from_file_location("ref_test_alphabet", os.path.join(REFERENCE, "tests", "test_alphabet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for arch in ("ESM-1b", "roberta_large"):  # esm1b_t33_650M_UR50S / esm1v_t33_650M_UR90S_* use this alphabet
        a = esm.Alphabet.from_architecture(arch)
        mod._test_esm1b(a)
        mod._test_esm1b_truncation(a)


def test_checkpoint_loading_matches_reference(probe):
    """esm.pretrained.load_model_and_alphabet on checkpoint FILES in the released formats (ESM-2 cfg format;
    ESM-1b / MSA Transformer fairseq-args formats with encoder_* argument names, encoder.sentence_encoder.* key
    prefixes, the row/column key swap of the MSA files, sibling -contact-regression.pt): same model class name, same
    alphabet, same state-dict keys, identical tensors as the reference's loader (esm/pretrained.py:52-221)."""
    import esm

    assert len(probe["checkpoints"]) == 3
    for c in probe["checkpoints"]:
        model, alphabet = esm.pretrained.load_model_and_alphabet(c["path"])
        assert type(model).__name__ == c["cls"], c["path"]
        assert list(alphabet.all_toks) == c["all_toks"] and model.num_layers == c["num_layers"]
        got = model.state_dict()
        assert sorted(got) == sorted(c["state"]), set(got) ^ set(c["state"])
        for k, v in c["state"].items():
            assert got[k].dtype == v.dtype and torch.equal(got[k], v), (c["path"], k)


def test_module_surface_matches_reference(probe):
    """What callers poke at on the nn.Module (scripts/extract.py:65-84, esmfold.py:61-67, the FSDP example wrapping
    ``model.layers``): child-module names and order, state-dict key order and shapes, parameter / buffer names,
    parameter count, the scalar attributes — identical to a freshly constructed reference ESM2 (esm2.py:15-75)."""
    import esm

    ref = probe["surface"]["esm2_8M"]
    m = esm.ESM2(6, 320, 20)
    assert [n for n, _ in m.named_children()] == ref["children"]
    assert list(m.state_dict().keys()) == ref["state_keys"]
    assert [n for n, _ in m.named_parameters()] == ref["param_names"]
    assert [n for n, _ in m.named_buffers()] == ref["buffer_names"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params"]
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref["shapes"]
    assert [n for n, _ in m.layers[0].named_children()] == ref["layer_children"]
    for a, v in ref["attrs"].items():
        assert getattr(m, a) == v, a
    d = esm.ESM2(num_layers=1)
    for a, v in probe["surface"]["esm2_default"]["attrs"].items():
        assert getattr(d, a) == v, a


def _check_surface(m, ref):
    assert [n for n, _ in m.named_children()] == ref["children"]
    assert list(m.state_dict().keys()) == ref["state_keys"]
    assert [n for n, _ in m.named_parameters()] == ref["param_names"]
    assert [n for n, _ in m.named_buffers()] == ref["buffer_names"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params"]
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref["shapes"]
    assert [n for n, _ in m.layers[0].named_children()] == ref["layer_children"]
    for a, v in ref["attrs"].items():
        assert getattr(m, a) == v, a


def test_msa_and_esm1b_module_surface_matches_reference(probe):
    """Same check for MSATransformer (msa_transformer.py:20-144) and ProteinBertModel / ESM-1b (esm1.py:20-115)."""
    import argparse

    import esm

    msa_args = argparse.Namespace(layers=2, embed_dim=96, ffn_embed_dim=192, attention_heads=3, dropout=0.1,
                                  attention_dropout=0.1, activation_dropout=0.1, max_positions=1024,
                                  embed_positions_msa=True, embed_positions_msa_dim=96, max_tokens=2 ** 14,
                                  max_tokens_per_msa=2 ** 14)
    _check_surface(esm.MSATransformer(msa_args, esm.Alphabet.from_architecture("msa_transformer")), probe["surface"]["msa"])
    b_args = argparse.Namespace(arch="roberta_large", layers=2, embed_dim=96, ffn_embed_dim=384, attention_heads=3,
                                max_positions=1024, token_dropout=True, emb_layer_norm_before=True)
    _check_surface(esm.ProteinBertModel(b_args, esm.Alphabet.from_architecture("roberta_large")), probe["surface"]["esm1b"])


def test_alphabet_object_and_dataset_edge_cases(probe, tmp_path):
    """Attributes and small methods of Alphabet for all five architectures the reference names (data.py:91-176), the
    error for an unknown architecture, FastaBatchedDataset indexing, the duplicate-label check of from_file
    (data.py:52-55) and a batch holding an empty string."""
    import esm

    for arch, ref in probe["alphabet_surface"].items():
        if arch == "_invalid":
            continue
        a = esm.Alphabet.from_architecture(arch)
        for f, v in ref["fields"].items():
            assert getattr(a, f) == v, (arch, f)
        assert len(a) == ref["length"] and a.to_dict() == ref["to_dict"]
        assert [(t, a.get_idx(t)) for t, _ in ref["get_idx"]] == ref["get_idx"], arch
        assert [(i, a.get_tok(i)) for i, _ in ref["get_tok"]] == ref["get_tok"], arch
        assert type(a.get_batch_converter()).__name__ == ref["converter"], arch
    with pytest.raises(Exception) as ei:
        esm.Alphabet.from_architecture("no_such_arch")
    assert type(ei.value).__name__ == probe["alphabet_surface"]["_invalid"]
    ds = esm.FastaBatchedDataset(["a", "b", "c"], ["MKT", "MKTVRQG", "M"])
    ref = probe["dataset"]
    assert len(ds) == ref["length"] and [ds[i] for i in range(3)] == ref["items"]
    assert ds.get_batch_indices(8, 1) == ref["batches"]
    f = tmp_path / "dup.fasta"
    f.write_text(">x\nMK\n>x\nMKT\n")
    with pytest.raises(Exception) as ei:
        esm.FastaBatchedDataset.from_file(f)
    assert (type(ei.value).__name__, str(ei.value)) == ref["duplicate_labels"]
    a = esm.Alphabet.from_architecture("ESM-1b")
    labels, strs, toks = a.get_batch_converter()([("e", ""), ("f", "MK")])
    assert (labels, strs) == ref["empty_string_batch"][:2] and torch.equal(toks, ref["empty_string_batch"][2])


def test_binding_stub_reads_the_real_reference_class(probe):
    """examples/reference_binding/_esmk.py configured from an instance of the REFERENCE's ESM2 (attribute names of
    esm2.py:24-38): esmk_create accepts it and plans the same packed image as for this repo's class."""
    import ctypes

    import esm
    from esm_amd import _native

    if "stub" not in probe:
        pytest.skip("libesmk.so not built")
    assert probe["stub"]["rc"] == 0
    m = esm.ESM2(6, 320, 20)
    cfg = _native.EsmkConfig(6, 320, 20, 1280, m.alphabet_size, m.padding_idx, m.mask_idx, m.cls_idx, m.eos_idx, 1, 1, 1,
                             _native.dtype_code(torch.float16), 0, 0, 0)
    assert probe["stub"]["cfg"] == [getattr(cfg, f[0]) for f in _native.EsmkConfig._fields_]
    h, n = ctypes.c_void_p(), ctypes.c_size_t()
    _native.check(_native.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)))
    _native.check(_native.lib.esmk_packed_bytes(h, ctypes.byref(n)))
    _native.lib.esmk_destroy(h)
    assert n.value == probe["stub"]["packed_bytes"]
    assert probe["stub"]["state_keys"] == list(m.state_dict())  # every key esmk_pack_weight will be handed


def test_extraction_driver_writes_the_files_the_reference_script_writes(tmp_path):
    """esm_amd.extract.extract (the sharded driver; here one rank on CPU with the oracle as its embed_fn — test
    infrastructure, the product's embed_fn is the engine) against the reference's scripts/extract.py run on the same
    checkpoint and FASTA: the same set of <label>.pt files, the same keys, dtypes and shapes, truncation at
    --truncation_seq_length, values to oracle precision."""
    import pathlib

    import esm
    from esm_amd.extract import extract
    from esm_amd.synth import synth_esm2_state_dict, write_esm2_checkpoint
    from oracle.esm2_oracle import esm2_forward

    L, E, H = 3, 128, 2
    ckpt = write_esm2_checkpoint(str(tmp_path), "esm2_t3_tiny_UR50D", L, E, H, seed=17)
    g = torch.Generator().manual_seed(8)
    aas = "LAGVSERTIDPKQNFYMHWC"
    seqs = {f"sp|P{i:05d}|NAME_{i} desc": "".join(aas[j] for j in torch.randint(0, 20, (n,), generator=g).tolist())
            for i, n in enumerate([12, 55, 31, 70, 8])}
    fasta = tmp_path / "in.fasta"
    fasta.write_text("".join(f">{k}\n{v}\n" for k, v in seqs.items()))
    common = ["--repr_layers", "-1", "1", "--include", "mean", "per_tok", "bos", "contacts", "--truncation_seq_length", "40",
              "--toks_per_batch", "150"]
    r = subprocess.run([sys.executable, os.path.join(REFERENCE, "scripts", "extract.py"), ckpt, str(fasta),
                        str(tmp_path / "ref"), "--nogpu"] + common, capture_output=True, text=True, cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=REFERENCE, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]

    sd = synth_esm2_state_dict(L, E, H, seed=17)
    alphabet = esm.Alphabet.from_architecture("ESM-1b")
    ds = esm.FastaBatchedDataset.from_file(fasta)

    def embed_fn(toks, layers, return_contacts):
        return esm2_forward(sd, toks, L, H, repr_layers=layers, return_contacts=return_contacts)

    extract(ds, alphabet, embed_fn, L, E, [-1, 1], ["mean", "per_tok", "bos", "contacts"],
            output_dir=pathlib.Path(tmp_path / "ours"), toks_per_batch=150, truncation_seq_length=40, device=None,
            gather_mean=False, log=lambda s: None)
    ref_files = sorted(p.relative_to(tmp_path / "ref") for p in (tmp_path / "ref").rglob("*.pt"))
    our_files = sorted(p.relative_to(tmp_path / "ours") for p in (tmp_path / "ours").rglob("*.pt"))
    assert ref_files == our_files and len(ref_files) == 5
    for rel in ref_files:
        a = torch.load(tmp_path / "ref" / rel, weights_only=False)
        b = torch.load(tmp_path / "ours" / rel, weights_only=False)
        assert a["label"] == b["label"] and sorted(a) == sorted(b)
        for key in ("representations", "mean_representations", "bos_representations"):
            assert sorted(a[key]) == sorted(b[key]) == [1, L]
            for l in a[key]:
                assert a[key][l].shape == b[key][l].shape and a[key][l].dtype == b[key][l].dtype, (rel, key, l)
                assert (a[key][l] - b[key][l]).abs().max().item() < 5e-5
        assert a["contacts"].shape == b["contacts"].shape and (a["contacts"] - b["contacts"]).abs().max().item() < 5e-5
