"""Runs INSIDE the reference implementation (sys.path -> /root/reference; never imported by the test process, whose
``esm`` is this repo's shim) and pickles what the reference computes for seeded random inputs.  Used by
tests/test_reference_differential.py; the synthetic weights come from esm_amd.synth (loaded by file path so that
``import esm`` keeps meaning the reference here).

    python tests/_reference_probe.py out.pkl
"""
import argparse
import importlib.util
import os
import pickle
import random
import sys

import torch

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_synth():
    spec = importlib.util.spec_from_file_location("esm_amd_synth", os.path.join(ROOT, "esm_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rand_seq(rng, n, alphabet="LAGVSERTIDPKQNFYMHWCXBUZO"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def main():
    out_path = sys.argv[1]
    sys.path.insert(0, REFERENCE)
    import esm  # the reference

    assert esm.__file__.startswith(REFERENCE), esm.__file__
    synth = load_synth()
    rng = random.Random(1234)
    res = {}

    # ---- tokeniser / batch converter (reference esm/data.py:91-297) ----------------------------------
    res["alphabet"] = {}
    for arch in ("ESM-1b", "roberta_large", "ESM-1", "msa_transformer"):
        a = esm.Alphabet.from_architecture(arch)
        strings = [rand_seq(rng, rng.randint(1, 60)) for _ in range(12)]
        strings += ["M-K.X", "<cls>MK<eos>", "", " ", "K A <mask> I S Q", "KALTA<mask>ISQP", "MK<pad>T", "<unk>MK", "M<null_1>K"]
        cases = dict(all_toks=list(a.all_toks), ids=(a.padding_idx, a.cls_idx, a.eos_idx, a.unk_idx, a.mask_idx),
                     bos_eos=(a.prepend_bos, a.append_eos), encode=[], tokenize=[], batches=[], raises=[])
        for s in strings:
            try:
                cases["encode"].append((s, a.encode(s)))
                cases["tokenize"].append((s, a.tokenize(s)))
            except KeyError:
                cases["raises"].append(s)
        for s in ("MKJ", "mk", "M*K"):
            try:
                a.encode(s)
            except KeyError:
                cases["raises"].append(s)
        if arch != "msa_transformer":
            for trunc in (None, 10, 1022):
                batch = [(f"id{i}", rand_seq(rng, rng.randint(1, 40), "LAGVSERTIDPKQNFYMHWC")) for i in range(5)]
                batch.append(("masked", "KALTA<mask>ISQPISQP"))
                labels, strs, toks = a.get_batch_converter(trunc)(batch)
                cases["batches"].append((trunc, batch, (list(labels), list(strs), toks)))
        res["alphabet"][arch] = cases

    a = esm.Alphabet.from_architecture("msa_transformer")
    conv = a.get_batch_converter()
    res["msa_batches"] = []
    for _ in range(3):
        width = rng.randint(3, 30)
        msa = [(f"r{i}", rand_seq(rng, width, "LAGVSERTIDPKQNFYMHWC-")) for i in range(rng.randint(1, 6))]
        labels, strs, toks = conv(msa)
        res["msa_batches"].append((msa, (labels, strs, toks)))
    two = [[(f"a{i}", rand_seq(rng, 9, "LAGV-")) for i in range(3)], [(f"b{i}", rand_seq(rng, 14, "KQNF-")) for i in range(5)]]
    labels, strs, toks = conv(two)
    res["msa_batches"].append((two, (labels, strs, toks)))

    # ---- FASTA dataset + token-budget batching (reference esm/data.py:24-88) -------------------------
    res["fasta"] = []
    example = open(os.path.join(REFERENCE, "examples", "data", "some_proteins.fasta")).read()
    texts = [example]
    for _ in range(3):
        n = rng.randint(1, 25)
        lines = []
        for i in range(n):
            s = rand_seq(rng, rng.randint(1, 400), "LAGVSERTIDPKQNFYMHWC")
            lines.append(f">seq{i} some description {i}")
            for j in range(0, len(s), 60):  # wrapped lines
                lines.append(s[j:j + 60])
        texts.append("\n".join(lines) + "\n")
    import tempfile

    for text in texts:
        with tempfile.NamedTemporaryFile("w", suffix=".fasta", delete=False) as f:
            f.write(text)
            name = f.name
        ds = esm.FastaBatchedDataset.from_file(name)
        for tpb, extra in ((4096, 1), (1024, 1), (300, 0), (10, 2)):
            res["fasta"].append((text, tpb, extra, (list(ds.sequence_labels), list(ds.sequence_strs),
                                                      ds.get_batch_indices(tpb, extra_toks_per_seq=extra))))
        os.unlink(name)
    res["read_fasta"] = []
    a3m = os.path.join(REFERENCE, "examples", "data", "1a3a_1_A.a3m")
    for kw in (dict(), dict(keep_gaps=False), dict(keep_insertions=False, to_upper=True)):
        res["read_fasta"].append((a3m, kw, list(esm.data.read_fasta(a3m, **kw))[:20]))

    # ---- model forwards on synthetic weights at dims / seeds outside the committed fixtures ----------
    def toks2d(B, T, seed, ragged=True):
        g = torch.Generator().manual_seed(seed)
        t = torch.randint(4, 24, (B, T), generator=g)
        t[:, 0] = 0
        t[:, -1] = 2
        if ragged and B > 1:
            cut = T - max(2, T // 3)
            t[1, cut] = 2
            t[1, cut + 1:] = 1
            t[0, 2] = 32
            t[0, T // 2] = 30
        return t

    res["esm2"] = []
    for L, E, H, seed, B, T in ((2, 480, 20, 101, 2, 37), (3, 640, 20, 102, 2, 23), (2, 256, 2, 103, 3, 41),
                                (4, 128, 2, 104, 1, 140), (1, 320, 20, 105, 2, 9)):
        sd = synth.synth_esm2_state_dict(L, E, H, seed=seed)
        m = esm.ESM2(num_layers=L, embed_dim=E, attention_heads=H, alphabet="ESM-1b", token_dropout=True).eval()
        m.load_state_dict(sd, strict=True)
        t = toks2d(B, T, seed)
        with torch.no_grad():
            o = m(t, repr_layers=list(range(L + 1)), return_contacts=True)
        res["esm2"].append(dict(L=L, E=E, H=H, seed=seed, tokens=t, logits=o["logits"].float(),
                                representations={k: v.float() for k, v in o["representations"].items()},
                                attentions=o["attentions"].float(), contacts=o["contacts"].float()))

    res["esm1b"] = []
    ab = esm.Alphabet.from_architecture("roberta_large")
    for L, E, H, seed, B, T, lnb in ((2, 256, 4, 111, 2, 33, True), (3, 128, 2, 112, 3, 18, False)):
        sd = synth.synth_esm1b_state_dict(L, E, H, seed=seed, ln_before=lnb)
        args = argparse.Namespace(arch="roberta_large", layers=L, embed_dim=E, ffn_embed_dim=4 * E, attention_heads=H,
                                  max_positions=1024, token_dropout=True, emb_layer_norm_before=lnb)
        m = esm.ProteinBertModel(args, ab).eval()
        m.load_state_dict(sd, strict=True)
        t = toks2d(B, T, seed)
        with torch.no_grad():
            o = m(t, repr_layers=list(range(L + 1)))
        res["esm1b"].append(dict(L=L, E=E, H=H, seed=seed, ln_before=lnb, tokens=t, logits=o["logits"].float(),
                                 representations={k: v.float() for k, v in o["representations"].items()}))

    res["msa"] = []
    am = esm.Alphabet.from_architecture("msa_transformer")
    for L, E, H, F, seed, B, R, C in ((2, 128, 2, 256, 121, 1, 7, 26), (1, 192, 3, 384, 122, 2, 4, 15)):
        sd = synth.synth_msa_state_dict(L, E, H, F, seed=seed)
        args = argparse.Namespace(layers=L, embed_dim=E, ffn_embed_dim=F, attention_heads=H, dropout=0.1,
                                  attention_dropout=0.1, activation_dropout=0.1, max_positions=1024,
                                  embed_positions_msa=True, embed_positions_msa_dim=E, max_tokens=2 ** 14,
                                  max_tokens_per_msa=2 ** 14)
        m = esm.MSATransformer(args, am).eval()
        m.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed)
        t = torch.randint(4, 24, (B, R, C), generator=g)
        t[torch.rand((B, R, C), generator=g) < 0.1] = 30
        t[:, :, 0] = 0
        if B > 1:
            t[1, :, C - 2:] = 1
            t[1, R - 1, :] = 1
        with torch.no_grad():
            o = m(t, repr_layers=list(range(L + 1)), return_contacts=True)
        res["msa"].append(dict(L=L, E=E, H=H, F=F, seed=seed, tokens=t, logits=o["logits"].float(),
                               representations={k: v.float() for k, v in o["representations"].items()},
                               row_attentions=o["row_attentions"].float(), contacts=o["contacts"].float()))

    # ---- checkpoint files in the released formats, loaded by the reference's esm.pretrained -----------------
    ck_dir = os.path.join(os.path.dirname(os.path.abspath(out_path)), "ckpt")
    os.makedirs(ck_dir, exist_ok=True)
    res["checkpoints"] = []

    def split_regression(sd):
        reg = {k: v for k, v in sd.items() if k.startswith("contact_head.")}
        body = {k: v for k, v in sd.items() if not k.startswith("contact_head.")}
        return body, reg

    def record(path):
        model, alphabet = esm.pretrained.load_model_and_alphabet(path)
        res["checkpoints"].append(dict(path=path, cls=type(model).__name__, all_toks=list(alphabet.all_toks),
                                       state={k: v.clone() for k, v in model.state_dict().items()},
                                       num_layers=model.num_layers))

    # ESM-2 (cfg format, esm/pretrained.py:164-188)
    record(synth.write_esm2_checkpoint(ck_dir, "esm2_t3_synth_UR50D", 3, 128, 2, seed=201))
    # ESM-1b (fairseq args format with encoder_* argument names and encoder.sentence_encoder.* keys, :87-99)
    body, reg = split_regression(synth.synth_esm1b_state_dict(2, 128, 2, seed=202, ln_before=True))
    args = argparse.Namespace(arch="roberta_large", encoder_layers=2, encoder_embed_dim=128, encoder_ffn_embed_dim=512,
                              encoder_attention_heads=2, max_positions=1024, token_dropout=True)
    path = os.path.join(ck_dir, "esm1b_t2_synth_UR50S.pt")
    torch.save({"args": args, "model": {"encoder.sentence_encoder." + k: v for k, v in body.items()}}, path)
    torch.save({"model": reg}, path[:-3] + "-contact-regression.pt")
    record(path)
    # MSA Transformer (args format; released files have row / column swapped in the key names, :109-121)
    body, reg = split_regression(synth.synth_msa_state_dict(2, 128, 2, 256, seed=203))
    swap = lambda s: s.replace("row", "column") if "row" in s else s.replace("column", "row")
    args = argparse.Namespace(arch="msa_transformer", encoder_layers=2, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
                              encoder_attention_heads=2, dropout=0.1, attention_dropout=0.1, activation_dropout=0.1,
                              max_positions=1024, embed_positions_msa=True, max_tokens=2 ** 14, max_tokens_per_msa=2 ** 14)
    path = os.path.join(ck_dir, "esm_msa1b_t2_synth_UR50S.pt")
    torch.save({"args": args, "model": {"encoder.sentence_encoder." + swap(k): v for k, v in body.items()}}, path)
    torch.save({"model": reg}, path[:-3] + "-contact-regression.pt")
    record(path)

    # ---- Alphabet object surface + dataset edge cases (esm/data.py:91-297, 24-88) -------------------------------
    res["alphabet_surface"] = {}
    for arch in ("ESM-1b", "roberta_large", "ESM-1", "msa_transformer", "invariant_gvp"):
        a = esm.Alphabet.from_architecture(arch)
        fields = ("standard_toks", "prepend_toks", "append_toks", "prepend_bos", "append_eos", "use_msa", "all_toks",
                  "tok_to_idx", "unk_idx", "padding_idx", "cls_idx", "mask_idx", "eos_idx", "all_special_tokens",
                  "unique_no_split_tokens")
        res["alphabet_surface"][arch] = dict(
            fields={f: getattr(a, f) for f in fields}, length=len(a), to_dict=a.to_dict(),
            get_idx=[(t, a.get_idx(t)) for t in ("L", "J", "<mask>", "<cls>", "-", "nonsense")],
            get_tok=[(i, a.get_tok(i)) for i in (0, 1, 5, len(a) - 1)],
            converter=type(a.get_batch_converter()).__name__)
    try:
        esm.Alphabet.from_architecture("no_such_arch")
        res["alphabet_surface"]["_invalid"] = None
    except Exception as e:  # noqa: BLE001
        res["alphabet_surface"]["_invalid"] = type(e).__name__
    ds = esm.FastaBatchedDataset(["a", "b", "c"], ["MKT", "MKTVRQG", "M"])
    res["dataset"] = dict(length=len(ds), items=[ds[i] for i in range(3)], batches=ds.get_batch_indices(8, 1))
    import tempfile as _tf

    with _tf.NamedTemporaryFile("w", suffix=".fasta", delete=False) as f:
        f.write(">x\nMK\n>x\nMKT\n")
        dup = f.name
    try:
        esm.FastaBatchedDataset.from_file(dup)
        res["dataset"]["duplicate_labels"] = None
    except Exception as e:  # noqa: BLE001
        res["dataset"]["duplicate_labels"] = (type(e).__name__, str(e))
    os.unlink(dup)
    a = esm.Alphabet.from_architecture("ESM-1b")
    labels, strs, toks = a.get_batch_converter()([("e", ""), ("f", "MK")])
    res["dataset"]["empty_string_batch"] = (labels, strs, toks)

    # ---- nn.Module surface of a freshly constructed model (SURVEY.md §8 b) --------------------------------------
    def surface(m, attrs):
        return dict(children=[n for n, _ in m.named_children()], state_keys=list(m.state_dict().keys()),
                    param_names=[n for n, _ in m.named_parameters()], buffer_names=[n for n, _ in m.named_buffers()],
                    n_params=sum(p.numel() for p in m.parameters()),
                    shapes={k: tuple(v.shape) for k, v in m.state_dict().items()},
                    attrs={a: getattr(m, a) for a in attrs},
                    layer_children=[n for n, _ in m.layers[0].named_children()])

    scalar_attrs = ["num_layers", "embed_dim", "attention_heads", "alphabet_size", "padding_idx", "mask_idx", "cls_idx",
                    "eos_idx", "prepend_bos", "append_eos", "token_dropout", "embed_scale"]
    msa_args = argparse.Namespace(layers=2, embed_dim=96, ffn_embed_dim=192, attention_heads=3, dropout=0.1,
                                  attention_dropout=0.1, activation_dropout=0.1, max_positions=1024,
                                  embed_positions_msa=True, embed_positions_msa_dim=96, max_tokens=2 ** 14,
                                  max_tokens_per_msa=2 ** 14)
    b_args = argparse.Namespace(arch="roberta_large", layers=2, embed_dim=96, ffn_embed_dim=384, attention_heads=3,
                                max_positions=1024, token_dropout=True, emb_layer_norm_before=True)
    res["surface"] = {
        "msa": surface(esm.MSATransformer(msa_args, esm.Alphabet.from_architecture("msa_transformer")),
                       ["num_layers", "padding_idx", "mask_idx", "cls_idx", "eos_idx", "prepend_bos", "append_eos",
                        "alphabet_size"]),
        "esm1b": surface(esm.ProteinBertModel(b_args, esm.Alphabet.from_architecture("roberta_large")),
                         ["num_layers", "padding_idx", "mask_idx", "cls_idx", "eos_idx", "prepend_bos", "append_eos",
                          "alphabet_size", "model_version"]),
        "esm2_8M": surface(esm.ESM2(6, 320, 20), scalar_attrs),
        "esm2_default": {"attrs": {a: getattr(esm.ESM2(num_layers=1), a) for a in ("embed_dim", "attention_heads")}},
    }

    # ---- the reference-side binding stub on the REAL reference class (examples/reference_binding/_esmk.py) ------
    import ctypes

    spec = importlib.util.spec_from_file_location("ref_esmk_stub", os.path.join(ROOT, "examples", "reference_binding", "_esmk.py"))
    stub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stub)
    libpath = os.path.join(ROOT, "esm_amd", "lib", "libesmk.so")
    if os.path.exists(libpath):
        L_ = stub.lib(libpath)
        m = esm.ESM2(6, 320, 20)
        cfg = stub.config_for(m)  # reads the attributes the reference's ESM2.__init__ sets
        h, n = ctypes.c_void_p(), ctypes.c_size_t()
        rc = L_.esmk_create(ctypes.byref(cfg), ctypes.byref(h))
        L_.esmk_packed_bytes(h, ctypes.byref(n))
        L_.esmk_destroy(h)
        res["stub"] = dict(rc=rc, packed_bytes=n.value, cfg=[getattr(cfg, f[0]) for f in stub.Config._fields_],
                           state_keys=[k for k in m.state_dict()])

    with open(out_path, "wb") as f:
        pickle.dump(res, f)


if __name__ == "__main__":
    main()
