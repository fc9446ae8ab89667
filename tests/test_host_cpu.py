"""Host-side logic on CPU: tokeniser / batcher golden vectors of the reference's own tests,
FASTA batching, checkpoint round trip, drop-in names, and the C ABI surface of libesmk.so."""
import os
import re

import pytest
import torch

import esm
from esm_amd import _native
from esm_amd.synth import synth_esm2_state_dict, write_esm2_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- reference tests/test_alphabet.py:6-24 ---------------------------------------------------
def test_alphabet_golden_tokens():
    alphabet = esm.Alphabet.from_architecture("ESM-1b")
    conv = alphabet.get_batch_converter()
    data = [("protein1", "MKTVRQG"), ("protein2 with mask", "KALTA<mask>ISQP"),
            ("protein3 with mask and spaces", "K A <mask> I S Q")]
    labels, strs, toks = conv(data)
    expected = torch.tensor([[0, 20, 15, 11, 7, 10, 16, 6, 2, 1, 1, 1],
                             [0, 15, 5, 4, 11, 5, 32, 12, 8, 16, 14, 2],
                             [0, 15, 5, 32, 12, 8, 16, 2, 1, 1, 1, 1]])
    assert torch.equal(toks, expected)
    assert labels == [d[0] for d in data] and strs == [d[1] for d in data]
    assert toks.dtype == torch.int64


# ---- reference tests/test_alphabet.py:27-45 --------------------------------------------------
def test_alphabet_truncation_golden():
    alphabet = esm.Alphabet.from_architecture("ESM-1b")
    conv = alphabet.get_batch_converter(truncation_seq_length=10)
    data = [("p1", "MKTVRQGMKTVRQG"), ("p2", "KALTA<mask>ISQPISQP"), ("p3", "K A <mask> I S Q")]
    _, strs, toks = conv(data)
    expected = torch.tensor([[0, 20, 15, 11, 7, 10, 16, 6, 20, 15, 11, 2],
                             [0, 15, 5, 4, 11, 5, 32, 12, 8, 16, 14, 2],
                             [0, 15, 5, 32, 12, 8, 16, 2, 1, 1, 1, 1]])
    assert torch.equal(toks, expected)
    assert strs[0] == "MKTVRQGMKTVRQG"  # strings stay untruncated


# ---- reference tests/test_alphabet.py:64-87 --------------------------------------------------
def test_msa_batch_converter_golden():
    alphabet = esm.Alphabet.from_architecture("msa_transformer")
    conv = alphabet.get_batch_converter()
    _, _, toks = conv([("1", "MKTVRQG"), ("2", "KALTRAI"), ("3", "KAAISQQ")])
    expected = torch.tensor([[[0, 20, 15, 11, 7, 10, 16, 6], [0, 15, 5, 4, 11, 10, 5, 12],
                              [0, 15, 5, 5, 12, 8, 16, 16]]])
    assert torch.equal(toks, expected)
    with pytest.raises(RuntimeError):
        conv([("1", "MKT"), ("2", "MK")])


def test_alphabet_ids_and_edge_cases():
    for arch, n, cls, mask in [("ESM-1b", 33, 0, 32), ("msa_transformer", 33, 0, 32), ("ESM-1", 35, 32, 33)]:
        a = esm.Alphabet.from_architecture(arch)
        assert (len(a), a.padding_idx, a.eos_idx, a.unk_idx, a.cls_idx, a.mask_idx) == (n, 1, 2, 3, cls, mask)
    a = esm.Alphabet.from_architecture("ESM-1b")
    assert a.all_toks[4:31] == list("LAGVSERTIDPKQNFYMHWCXBUZO.-") and a.all_toks[31] == "<null_1>"
    assert a.encode("M-K.X") == [20, 30, 15, 29, 24]
    assert a.encode("<cls>MK<eos>") == [0, 20, 15, 2]
    assert a.encode("") == [] and a.encode(" ") == []
    assert a.tokenize("K A <mask> I S Q") == a.tokenize("KA<mask>ISQ")
    for bad in ("MKJ", "mkt", "*"):
        with pytest.raises(KeyError):
            a.encode(bad)
    with pytest.raises(ValueError):
        esm.Alphabet.from_architecture("nope")
    _, strs, toks = a.get_batch_converter(1022)([("a", "A" * 5000)])
    assert toks.shape == (1, 1024) and toks[0, 0] == 0 and toks[0, -1] == 2 and len(strs[0]) == 5000


def test_fasta_dataset_and_batching(tmp_path):
    lengths = [428, 222, 502, 156, 80, 71, 99, 602, 127, 152, 98, 935, 102, 882, 659]  # some_proteins.fasta
    fa = tmp_path / "x.fasta"
    with open(fa, "w") as f:
        for i, n in enumerate(lengths):
            f.write(f">seq{i} desc\n")
            s = "ACDEFGHIKL" * (n // 10 + 1)
            s = s[:n]
            for j in range(0, n, 60):
                f.write(s[j:j + 60] + "\n")
            f.write("\n")
    ds = esm.FastaBatchedDataset.from_file(fa)
    assert len(ds) == 15 and [len(s) for s in ds.sequence_strs] == lengths and ds[0][0] == "seq0 desc"
    # golden values measured on the reference (SURVEY.md §8 c)
    assert ds.get_batch_indices(4096, 1) == [[5, 4, 10, 6, 12, 8, 9, 3, 1], [0, 2, 7, 14], [13, 11]]
    assert ds.get_batch_indices(1024, 1) == [[5, 4, 10, 6, 12, 8], [9, 3, 1], [0, 2], [7], [14], [13], [11]]
    fb = tmp_path / "y.fasta"
    fb.write_text(">\nMK\n>a\nTT\n")
    assert esm.FastaBatchedDataset.from_file(fb).sequence_labels == ["seqnum000000000", "a"]
    fc = tmp_path / "z.fasta"
    fc.write_text(">a\nMK\n>a\nTT\n")
    with pytest.raises(AssertionError):
        esm.FastaBatchedDataset.from_file(fc)
    from esm.data import read_fasta

    fd = tmp_path / "w.a3m"
    fd.write_text(">q\nMK-aT\n>r x\nM.KtT\n")
    assert list(read_fasta(fd)) == [("q", "MK-aT"), ("r x", "M.KtT")]
    assert list(read_fasta(fd, keep_gaps=False, keep_insertions=False, to_upper=True)) == [("q", "MKT"), ("r x", "M.KT")]


def test_checkpoint_round_trip(tmp_path):
    path = write_esm2_checkpoint(str(tmp_path), "esm2_t2_tiny_UR50D", 2, 128, 2, seed=5)
    model, alphabet = esm.pretrained.load_model_and_alphabet(path)
    assert isinstance(model, esm.ESM2) and len(alphabet) == 33
    assert (model.num_layers, model.embed_dim, model.attention_heads) == (2, 128, 2)
    sd = synth_esm2_state_dict(2, 128, 2, seed=5)
    got = model.state_dict()
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert got["lm_head.weight"].data_ptr() == got["embed_tokens.weight"].data_ptr()  # tied
    # no regression file -> allowed only for names flagged as such
    os.remove(str(tmp_path / "esm2_t2_tiny_UR50D-contact-regression.pt"))
    with pytest.raises(FileNotFoundError):
        esm.pretrained.load_model_and_alphabet(path)
    os.rename(path, str(tmp_path / "esm2_t2_tiny_500K_UR50D.pt"))
    with pytest.warns(UserWarning):
        esm.pretrained.load_model_and_alphabet(str(tmp_path / "esm2_t2_tiny_500K_UR50D.pt"))


def test_state_dict_key_count_8m():
    m = esm.ESM2(6, 320, 20)
    assert len(m.state_dict()) == 113  # SURVEY.md §8 b: 113 tensors for the 8M model
    assert hasattr(esm.pretrained, "esm2_t33_650M_UR50D") and hasattr(esm.pretrained, "esm2_t36_3B_UR50D")


def test_forward_refuses_cpu_tensors():
    m = esm.ESM2(1, 128, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.tensor([[0, 5, 2]]))


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "esmk.h")).read()
    declared = set(re.findall(r"\b(esmk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    for name in declared:
        assert hasattr(_native.lib, name)
    assert b"gfx950" in _native.lib.esmk_version()


def test_library_carries_the_hash_of_its_sources():
    """esmk_version() ends in the SHA-256 prefix of the sources the .so was compiled from (esm_amd/build.py); the
    build is gated on it (not on mtimes) and bench.py uses it to decide whether the committed PMC traffic numbers
    belong to the running binary."""
    from esm_amd import build

    version = _native.lib.esmk_version().decode()
    assert version.endswith("esmk-src:" + build.source_hash()), (version, build.source_hash())
    assert build.library_hash() == build.source_hash() and not build.needs_build()


def test_self_launch_helpers():
    from esm_amd import launch

    p = launch.free_port()
    assert 1024 < p < 65536
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE") if k in os.environ}
    try:
        assert not launch.under_launcher()
        assert launch.init_ranks(1, "gloo") == (None, 0, 1, 0)  # no launcher, one rank: no process group
        with pytest.raises(SystemExit):
            launch.init_ranks(2, "gloo")                         # asked for 2 ranks, launched as 1
    finally:
        os.environ.update(env)


def test_pack_plan_layout():
    """Host side of token-packed batches (esm_amd/packing.py): segment starts on multiples of 16, the row count a
    multiple of 128, pack / unpack are inverse on the non-pad positions."""
    from esm_amd.packing import pack_plan

    toks = torch.tensor([[0, 5, 6, 2, 1, 1], [0, 5, 2, 1, 1, 1], [0, 4, 1, 4, 4, 2]])  # row 2: interior <pad>
    plan = pack_plan(toks, 1)
    assert plan.lengths.tolist() == [4, 3, 6] and plan.rows == 128
    assert plan.segments.tolist() == [[0, 4], [16, 3], [32, 6]] and plan.segments.dtype == torch.int32
    idx, keep = plan.index("cpu")
    flat = plan.pack(toks, 1, idx)
    assert flat.shape == (128,)
    assert flat[:4].tolist() == [0, 5, 6, 2] and flat[16:19].tolist() == [0, 5, 2] and flat[32:38].tolist() == [0, 4, 1, 4, 4, 2]
    assert int(flat.ne(1).sum()) == int(toks.ne(1).sum())
    back = plan.unpack(flat.unsqueeze(1).float(), idx, keep).squeeze(-1).long()
    assert torch.equal(keep, torch.arange(6).unsqueeze(0) < plan.lengths.unsqueeze(1))
    assert torch.equal(back[keep], toks[keep]) and int(back[~keep].abs().sum()) == 0
    assert pack_plan(toks, 1, lengths=[6, 6, 6]).segments.tolist() == [[0, 6], [16, 6], [32, 6]]


def test_predict_contacts_refuses_cpu_tensors():
    """predict_contacts (reference esm2.py:146-147) takes the contacts-only engine path; like forward it has no CPU
    fallback."""
    m = esm.ESM2(1, 128, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.predict_contacts(torch.tensor([[0, 5, 6, 2]]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.tensor([[0, 5, 6, 2]]), contacts_only=True)


def test_native_lowp_rule(monkeypatch):
    """Model-dtype outputs are written by the engine only when the model dtype IS the operand dtype."""
    from esm_amd.esm2 import _native_lowp, _operand_dtype_for

    monkeypatch.delenv("ESM_AMD_NATIVE_LOWP", raising=False)
    monkeypatch.delenv("ESM_AMD_OPERAND", raising=False)
    assert _operand_dtype_for(torch.float32) == torch.float16 and _operand_dtype_for(torch.bfloat16) == torch.bfloat16
    assert _native_lowp(torch.float16, torch.float16) and _native_lowp(torch.bfloat16, torch.bfloat16)
    assert not _native_lowp(torch.float32, torch.float16)   # fp32 model: fp32 outputs
    assert not _native_lowp(torch.float16, torch.bfloat16)  # .half() model forced to bf16 operands: cast outside
    monkeypatch.setenv("ESM_AMD_NATIVE_LOWP", "0")
    assert not _native_lowp(torch.float16, torch.float16)


def test_extract_embed_fn_dispatch():
    """esm_amd.extract.make_embed_fn: token-packed forward without contacts, the contacts-only forward (no attention
    tensor) with contacts, the reference-shaped call for models that have neither (MSA Transformer)."""
    from esm_amd.extract import make_embed_fn

    calls = []

    class Engine:
        supports_varlen = True
        supports_contacts_only = True

        def forward_varlen(self, toks, repr_layers, lengths=None):
            calls.append(("varlen", tuple(repr_layers), lengths))
            return {}

        def __call__(self, toks, repr_layers, return_contacts=False, contacts_only=False):
            calls.append(("forward", tuple(repr_layers), return_contacts, contacts_only))
            return {}

    class Plain:
        def __call__(self, toks, repr_layers, return_contacts=False):
            calls.append(("plain", tuple(repr_layers), return_contacts))
            return {}

    toks = torch.zeros((2, 5), dtype=torch.int64)
    fn = make_embed_fn(Engine())
    assert fn.wants_lengths
    fn(toks, [3], False, lengths=[5, 4])
    fn(toks, [3], True, lengths=[5, 4])
    make_embed_fn(Engine(), varlen=False)(toks, [3], False)
    make_embed_fn(Plain())(toks, [1], True)
    assert calls == [("varlen", (3,), [5, 4]), ("forward", (3,), False, True), ("forward", (3,), False, False),
                     ("plain", (1,), True)]


def test_gelu_polynomial_of_the_epilogues_matches_erf():
    """The GELU of the GEMM epilogues (esm_amd/csrc/common.h: gelu_fast) is x (0.5 + u Q(t)); replay that
    evaluation in emulated fp32 (one rounding per FMA) and hold it against float64 erf (reference esm/modules.py:17-24).
    Two coefficient sets: fp32 outputs (LM head dense layer) degree 11 / clamp 4.75 — 2e-6 absolute inside the clamp,
    2e-6 |x| beyond; operand-dtype outputs (fc1, rounded to fp16 / bf16 in the same epilogue) degree 8 / clamp 4 —
    8e-6 absolute inside, 3.2e-5 |x| beyond (1 - Phi(4)).  RELATIVE to the value: <= 2.5e-4 (fp16's half ulp)
    wherever |gelu| > 0.03, <= 3e-5 above 0.25; the negative tail is a one-signed residue of at most 3.2e-5 |x|."""
    import sys

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fit_gelu_poly as fg

    src = open(os.path.join(root, "esm_amd", "csrc", "common.h")).read()

    def coefs(macro):
        body = re.search(r"#define " + macro + r"\s*\\\s*\{(.*?)\}", src, re.S).group(1).replace("\\", " ")
        return np.array([float(v.rstrip("f")) for v in body.replace("\n", " ").split(",")], dtype=np.float32)

    for macro, cname, kname, size, b_in, b_out in (("ESMK_GELU_COEF", "kGeluClamp", "kGeluK2", 12, 2e-6, 2e-6),
                                                    ("ESMK_GELU_COEF_T", "kGeluClampT", "kGeluK2T", 9, 8e-6, 3.2e-5)):
        coef = coefs(macro)
        clamp = float(re.search(cname + r" = ([0-9.]+)f", src).group(1))
        k2 = float(re.search(kname + r" = ([0-9.e+-]+)f", src).group(1))
        assert coef.size == size and np.float32(k2) == np.float32(2 / clamp**2)
        e_in, e_out = fg.report(coef, clamp)
        assert e_in < b_in and e_out < b_out, (macro, e_in, e_out)
        # the tails: beyond the clamp the value is x * (0.5 + u Q(1)) with u = +-clamp — x * (1 + eps) on the
        # right, x * eps on the left: it does not decay to 0 like the exact GELU, it stays an eps |x| residue of either
        # sign (degree 11: 7e-5 at x = -50, below fp16's smallest normal; degree 8: 1.6e-3 at x = -50 — nothing in a
        # trained ESM-2 sends fc1 outputs there).  Pinned here so that a coefficient change cannot silently widen it.
        xs = np.concatenate([np.linspace(-100.0, -clamp, 200001), np.linspace(clamp, 100.0, 200001)])
        got = fg.gelu_poly_f32(xs, coef, clamp).astype(np.float64)
        exact = np.where(xs > 0, xs, 0.0)  # gelu(x) to 1e-6 |x| (3.2e-5 |x|) beyond the clamp
        assert (np.abs(got - exact) / np.abs(xs)).max() < b_out
        assert np.abs(got[xs < 0]).max() < 100 * b_out
        # relative error inside the clamp, against float64 erf (the operand-dtype set is NOT "8 x below the half ulp" on small
        # values: 2.2e-4 at |gelu| ~ 0.03), and the sign / size of the negative tail's residue
        from scipy.special import erf

        xi = np.linspace(-clamp, clamp, 800001)
        gi = fg.gelu_poly_f32(xi, coef, clamp).astype(np.float64)
        ei = xi * 0.5 * (1.0 + erf(xi / np.sqrt(2.0)))
        relerr = np.abs(gi - ei) / np.maximum(np.abs(ei), 1e-30)
        assert relerr[np.abs(ei) > 0.03].max() < (2.5e-4 if size == 9 else 5e-5), (macro, relerr[np.abs(ei) > 0.03].max())
        assert relerr[np.abs(ei) > 0.25].max() < (3e-5 if size == 9 else 6e-6), (macro, relerr[np.abs(ei) > 0.25].max())
        xt = np.linspace(-10.0, -clamp, 20001)
        gt = fg.gelu_poly_f32(xt, coef, clamp).astype(np.float64)
        assert (gt <= 1e-7).all() and (np.abs(gt) <= (3.3e-5 if size == 9 else 1.5e-6) * np.abs(xt)).all(), macro


def test_precision_study_tool_floor_is_ordered():
    """tools/esm2_precision_study.py: the emulated fp16-operand forward differs from the fp32 one by the
    order of the operand precision, bf16 is ~8x worse than fp16, and no injection at all is exact."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import esm2_precision_study as ps
    from esm_amd.synth import synth_tokens

    L, E, H = 3, 128, 2
    sd = {k: v.float() for k, v in synth_esm2_state_dict(L, E, H, seed=5).items()}
    toks = synth_tokens(2, 40, seed=6)
    every = ("W", "A", "QK", "V", "P")
    with torch.no_grad():
        ref = ps.forward(sd, toks, L, H, (), torch.float16)
        assert torch.equal(ref, ps.forward(sd, toks, L, H, (), torch.bfloat16))
        rel = lambda t: ((t - ref).norm() / ref.norm()).item()
        f16, bf16 = rel(ps.forward(sd, toks, L, H, every, torch.float16)), rel(ps.forward(sd, toks, L, H, every, torch.bfloat16))
        w_only = rel(ps.forward(sd, toks, L, H, ("W",), torch.float16))
    assert 5e-5 < f16 < 2e-3 and 4 < bf16 / f16 < 16 and w_only < f16
    # and it is the oracle's forward: same result as oracle.esm2_oracle on the same inputs
    from oracle.esm2_oracle import esm2_forward

    want = esm2_forward(sd, toks, L, H, repr_layers=[L])["representations"][L]
    assert (ref - want).abs().max().item() < 1e-5


def test_pmc_summary_tells_the_two_residual_gemms_apart(tmp_path):
    """tools/pmc_summary.py turns rocprofv3 counter rows into per-class HBM bytes that bench.py reports as
    `roofline.traffic`.  The out projection and fc2 run the same kernel symbol (residual epilogue): they are told apart
    by launch order inside a layer (round 3: both on gemm9; round 3a: fc2 on gemm9, out_proj on gemm8)."""
    import json
    import sqlite3
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def make(path, names):
        c = sqlite3.connect(path)
        c.execute("create table counters_collection (dispatch_id int, kernel_name text, counter_name text, value real, duration real)")
        for i, (name, fetch) in enumerate(names):
            c.execute("insert into counters_collection values (?,?,?,?,?)", (i, name, "FETCH_SIZE", fetch, 1000.0 * (i + 1)))
            c.execute("insert into counters_collection values (?,?,?,?,?)", (i, name, "WRITE_SIZE", 10.0, 1000.0 * (i + 1)))
        c.commit()
        c.close()

    g9r = "_ZN4esmk12gemm9_kernelIDF16_Li4ELi0ELb0EEEvNS_8GemmArgsEPy"
    g8r = "_ZN4esmk12gemm8_kernelIDF16_Li4ELi0ELi0ELi0ELb0ELb0EEEvNS_8GemmArgsEPy"
    g9g = "_ZN4esmk12gemm9_kernelIDF16_Li2ELi0ELb0EEEvNS_8GemmArgsEPy"
    for tag, layer, want in (("same", [(g9r, 100.0), (g9g, 7.0), (g9r, 300.0)], (100.0, 300.0)),
                             ("split", [(g8r, 100.0), (g9g, 7.0), (g9r, 300.0)], (100.0, 300.0))):
        db, out = tmp_path / f"{tag}.db", tmp_path / f"{tag}.json"
        make(db, layer * 3)
        subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(out), str(db)], check=True,
                       capture_output=True)
        k = json.load(open(out))["kernels"]
        # (launches_profiled counts counter rows: 3 launches x the two counters of this synthetic pass)
        assert k["gemm_out_proj"]["launches_profiled"] == 6 and k["gemm_fc2"]["launches_profiled"] == 6, tag
        assert k["gemm_out_proj"]["fetch_kib"] == want[0] and k["gemm_fc2"]["fetch_kib"] == want[1], tag
        assert k["gemm_fc1_gelu"]["fetch_kib"] == 7.0


def test_environment_switches_of_the_python_engine(monkeypatch):
    """ESM_AMD_OPERAND -> esmk_config.weight_split (0 off, 1 f16x2, 2 f16x2a, 3 f16x2v; round 6) and operand dtype;
    ESM_AMD_DUAL_STREAM -> the row windows of the two-stream forward (default: the measured windows, the headline batch and
    B = 6 outside; "0" off; "lo:hi[,lo:hi]" explicit)."""
    from esm_amd import esm2

    for env, want in (("", 0), ("f16", 0), ("bf16", 0), ("f16x2", 1), ("fp16x2", 1), ("f16x2a", 2), ("F16X2A", 2), ("f16x2v", 3), ("f16x3", 4)):
        monkeypatch.setenv("ESM_AMD_OPERAND", env)
        assert esm2._weight_split() == want, env
        if want:
            assert esm2._operand_dtype_for(torch.float32) == torch.float16
    monkeypatch.delenv("ESM_AMD_OPERAND")
    monkeypatch.delenv("ESM_AMD_DUAL_STREAM", raising=False)
    on = [b for b in range(1, 130) if esm2._dual_stream_wanted(b * 1024)]
    assert 4 in on and 8 in on and 16 in on and 32 in on and 48 in on
    assert not {1, 2, 6, 64, 128} & set(on)
    monkeypatch.setenv("ESM_AMD_DUAL_STREAM", "0")
    assert esm2._dual_stream_window() is None and not esm2._dual_stream_wanted(8192)
    monkeypatch.setenv("ESM_AMD_DUAL_STREAM", "100:200,1000:2000")
    assert esm2._dual_stream_window() == [(100, 200), (1000, 2000)]
    assert esm2._dual_stream_wanted(150) and esm2._dual_stream_wanted(2000) and not esm2._dual_stream_wanted(500)


def test_ln_fold_is_not_chosen_for_checkpoints_with_small_layernorm_gains(monkeypatch):
    """ESM2._fold_setting (round 6): ESM_AMD_LN_FOLD wins when set; unset, the LayerNorm gains decide — esmk_config.ln_fold 0
    (library default: fold on) for the plain synthetic weights and for the stress set with outliers 200 x the stream (gain
    ratio 133: fold / plain floor 1.1, profiles/r6_outlier_stress_study.log), -1 (fold off) from gain ratio ~ 200 on (2000 x:
    fold floor 3.3 x the plain one)."""
    import esm
    from esm_amd import esm2
    from esm_amd.synth import add_outlier_channels, skip_param_init, synth_esm2_state_dict

    monkeypatch.delenv("ESM_AMD_LN_FOLD", raising=False)
    L, E, H = 33, 320, 20  # a quarter of the 650M width: the same h at a quarter of the outlier magnitude
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    want = {0.0: (0, 0.0, 0.0), 50.0: (0, 0.3, 0.5), 500.0: (-1, 3.0, 5.0), 5000.0: (-1, 30.0, 50.0)}
    for mag, (setting, lo, hi) in want.items():
        sd = synth_esm2_state_dict(L, E, H, seed=0)
        if mag:
            add_outlier_channels(sd, L, E, magnitude=mag)
        model.load_state_dict(sd)
        model.refresh_engine()  # (on a GPU the engine's parameter fingerprint does this: ESM2._engine_ready)
        assert model._fold_setting() == setting, mag
        h = model._fold_hazard
        assert lo <= h <= hi, (mag, h)
    monkeypatch.setenv("ESM_AMD_LN_FOLD", "1")
    assert model._fold_setting() == 1
    monkeypatch.setenv("ESM_AMD_LN_FOLD", "0")
    assert model._fold_setting() == -1
    # a zero gain (a pruned channel) is a small gain
    g = torch.ones(2, 64)
    assert esm2.ln_fold_hazard(g) == 0.0
    g[1, 5] = 0.0
    assert esm2.ln_fold_hazard(g) > 1e6
