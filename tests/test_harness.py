"""CPU: the host side of the path (plantcaduceus_amd/zero_shot.py, embeddings.py) against golden vectors
produced by the REFERENCE's own functions (oracle/gen_golden.py imports src/zero_shot_score.py and
src/train_XGBoost.py in the build container).  The model in the loop is the CPU oracle stand-in — test
infrastructure; the product model class has no CPU path."""
import json
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import caduceus_oracle as O
from plantcaduceus_amd import embeddings, zero_shot
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer


@pytest.fixture(scope="module")
def snp_df(golden_dir):
    return pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")


def _oracle_model(g):
    cfg = make_config("x", d_model=int(g["model_d_model"]), n_layer=int(g["model_n_layer"]))
    sd = synthetic_state_dict(cfg, seed=int(g["model_seed"]))
    return O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))


def test_example_table_integrity(snp_df):
    assert len(snp_df) == 190
    assert set(snp_df.columns) >= {"chr", "start", "end", "pos", "ref", "alt", "sequences"}
    assert (snp_df["sequences"].str.len() == 512).all()
    ok = snp_df["ref"].isin(list("ACGT")) & snp_df["alt"].isin(list("ACGT"))
    assert int(ok.sum()) == 185
    assert all(s[255] == r for s, r in zip(snp_df["sequences"], snp_df["ref"]))


def test_zero_shot_score_matches_reference_function(golden_dir, snp_df):
    g = np.load(os.path.join(golden_dir, "harness_zero_shot.npz"))
    df = snp_df[g["keep_mask"]]
    got = np.asarray(zero_shot.zero_shot_score(df, g["dirichlet_probs"]), dtype=np.float64)
    np.testing.assert_allclose(got, g["dirichlet_scores"], rtol=1e-6, atol=1e-7)


def test_extract_logits_matches_reference_loop(golden_dir, snp_df):
    g = np.load(os.path.join(golden_dir, "harness_zero_shot.npz"))
    df = snp_df[g["keep_mask"]].iloc[: int(g["model_rows"])]
    tok = CaduceusTokenizer()
    probs = zero_shot.extract_logits(_oracle_model(g), df["sequences"].tolist(), "cpu", int(g["token_idx"]), tok,
                                     batch_size=3)
    assert probs.shape == (int(g["model_rows"]), 4) and probs.dtype == np.float32
    np.testing.assert_allclose(probs, g["model_probs"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(np.asarray(zero_shot.zero_shot_score(df, probs)), g["model_scores"], rtol=1e-4, atol=1e-5)


def test_extract_embeddings_matches_reference_loop(golden_dir, snp_df):
    g = np.load(os.path.join(golden_dir, "harness_embeddings.npz"))
    z = np.load(os.path.join(golden_dir, "harness_zero_shot.npz"))
    seqs = snp_df[z["keep_mask"]]["sequences"].tolist()[: int(g["model_rows"])]
    emb = embeddings.extract_embeddings(_oracle_model(g), seqs, "cpu", int(g["token_idx"]), CaduceusTokenizer(), batch_size=4)
    assert emb.shape == g["embeddings"].shape and emb.dtype == np.float32
    np.testing.assert_allclose(emb, g["embeddings"], rtol=1e-5, atol=1e-6)


def test_tokenize_masked_equals_per_sequence_reference_semantics(snp_df):
    tok = CaduceusTokenizer()
    seqs = snp_df["sequences"].tolist()[:5] + ["acgtn" * 102 + "NN"]
    ids = zero_shot.tokenize_masked(seqs, tok, 255)
    assert ids.shape == (6, 512) and (ids[:, 255] == tok.mask_token_id).all()
    v = tok.get_vocab()
    for s, row in zip(seqs, ids):
        want = [v.get(ch.lower(), v["[UNK]"]) for ch in s]
        want[255] = tok.mask_token_id
        assert row.tolist() == want
    assert tok.encode_plus("ACGTNacgt", return_tensors="pt")["input_ids"].tolist() == [[3, 4, 5, 6, 2, 3, 4, 5, 6]]
    assert tok.mask_token_id == 1 and tok.encode_batch([]).shape == (0, 0)
    with pytest.raises(ValueError):
        tok.encode_batch(["ACG", "AC"])


def test_windows_match_reference_seq_from_vcf(golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "harness_windows.json")) as f:
        g = json.load(f)
    fa = tmp_path / "g.fa"
    with open(fa, "w") as f:
        for name, seq in g["genome"].items():
            f.write(f">{name} some description\n")
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
    vcf = tmp_path / "v.vcf"
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for c, p, r, alts in g["records"]:
            f.write(f"{c}\t{p}\t.\t{r}\t{','.join(alts)}\t.\tPASS\tDP=3\n")
    for tidx, want in g["windows"].items():
        args = types.SimpleNamespace(inputVCF=str(vcf), inputFasta=str(fa), tokenIdx=int(tidx))
        seqs, ridx = zero_shot.seq_from_vcf(args)
        assert ridx == want["record_indices"]
        assert seqs == want["sequences"]
        assert all(len(s) == 512 for s in seqs)
    # VCF writer: per-ALT scores, "." for non-SNV alts, only scored records written, header passed through
    args = types.SimpleNamespace(inputVCF=str(vcf), inputFasta=str(fa), tokenIdx=255, output=str(tmp_path / "o.vcf"))
    seqs, ridx = zero_shot.seq_from_vcf(args)
    probs = np.random.default_rng(0).dirichlet(np.ones(4), size=len(seqs))
    zero_shot.zero_shot_score_vcf(args, ridx, probs)
    lines = [l for l in open(args.output).read().splitlines() if not l.startswith("#")]
    assert len(lines) == len(ridx)
    recs = [g["records"][i] for i in ridx]
    for line, (c, p, r, alts), pr in zip(lines, recs, probs):
        info = line.split("\t")[7]
        assert info.startswith("DP=3;plantCAD_zero_shot=")
        vals = info.split("plantCAD_zero_shot=")[1].split(",")
        assert len(vals) == len(alts)
        for a, v in zip(alts, vals):
            if len(a) == 1 and a in "ACGT":
                assert abs(float(v) - np.log(pr["ACGT".index(a)] / pr["ACGT".index(r)])) < 1e-9
            else:
                assert v == "."


def test_cli_table_and_bed_outputs(golden_dir, tmp_path, monkeypatch):
    """config-1 plumbing: TSV in -> TSV/BED out through main(), with the oracle stand-in as the loaded model."""
    g = np.load(os.path.join(golden_dir, "harness_zero_shot.npz"))
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:12]
    inp = tmp_path / "in.tsv"
    src.to_csv(inp, sep="\t", index=False)
    model = _oracle_model(g)
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (model, CaduceusTokenizer()))
    out = tmp_path / "out.tsv"
    zero_shot.main(["-input-table", str(inp), "-output", str(out), "-model", "unused", "-device", "cpu", "-batchSize", "5"])
    res = pd.read_csv(out, delimiter="\t")
    ok = src["ref"].isin(list("ACGT")) & src["alt"].isin(list("ACGT"))
    assert len(res) == int(ok.sum()) and "zeroShotScore" in res.columns
    n = min(len(res), int(g["model_rows"]))
    np.testing.assert_allclose(res["zeroShotScore"].to_numpy()[:n], g["model_scores"][:n], rtol=1e-4, atol=1e-5)
    bed = tmp_path / "out.bed"
    zero_shot.main(["-input-table", str(inp), "-output", str(bed), "-model", "unused", "-device", "cpu", "-outBED"])
    b = pd.read_csv(bed, delimiter="\t", header=None)
    assert b.shape[1] == 6 and (b[2] - b[1] == 1).all()
    with pytest.raises(SystemExit):
        zero_shot.parse_args(["-input-vcf", "x.vcf", "-model", "m"])


def test_plantcad2_probability_helpers_match_reference(golden_dir, snp_df):
    """masked_probs / unmasked_probs vs the reference's own _masked_probs / _unmasked_probs (src/zero-shot-eval.py)."""
    from plantcaduceus_amd import plantcad2_eval as pe
    g = np.load(os.path.join(golden_dir, "harness_plantcad2.npz"))
    seqs = snp_df["sequences"].tolist()[: int(g["rows"])]
    model, tok = _oracle_model(g), CaduceusTokenizer()
    np.testing.assert_allclose(pe.masked_probs(model, tok, seqs, int(g["single_idx"]), "cpu", batch_size=4), g["single"], rtol=2e-5, atol=1e-7)
    multi = pe.masked_probs(model, tok, seqs, g["multi_idx"].tolist(), "cpu", batch_size=5)
    assert multi.shape == (len(seqs) * 3, 4)
    np.testing.assert_allclose(multi, g["multi"], rtol=2e-5, atol=1e-7)
    un = pe.unmasked_probs(seqs, tok, model, "cpu", batch_size=4)
    assert un.shape == g["unmasked"].shape == (len(seqs), 512, 4)
    np.testing.assert_allclose(un, g["unmasked"], rtol=2e-5, atol=1e-7)
    with pytest.raises(ValueError):
        pe.unmasked_probs(["ACGT", "ACG"], tok, model, "cpu")


def test_plantcad2_task_metrics_match_reference(golden_dir, tmp_path, capsys):
    """token / motif accuracy, AUROC, AUPRC, ref-prob and true-prob scores, SV boundary LLR: outputs of the reference's
    src/zero-shot-eval.py:181-320 (sklearn roc_curve / average_precision_score where it uses them) on seeded tables with
    lower-case and N bases, tied scores and a zero probability."""
    import json
    import pandas as pd
    from plantcaduceus_amd import plantcad2_eval as pe
    g = np.load(os.path.join(golden_dir, "harness_plantcad2_metrics.npz"))
    tab = json.load(open(os.path.join(golden_dir, "harness_plantcad2_metrics.json")))
    seqs, labels = tab["sequences"], g["labels"]
    pos, ml, ti = [int(x) for x in g["positions"]], int(g["motif_len"]), int(g["token_idx"])
    tt = pe.true_tokens(seqs, pos)
    assert list(tt) == list(g["true_tokens"])
    assert pe.token_accuracy(g["probs3"], tt) == pytest.approx(float(g["token_acc"]), abs=1e-12)
    assert pe.motif_accuracy(g["probs3"], tt, ml) == pytest.approx(float(g["motif_acc"]), abs=1e-12)
    assert 0.2 < float(g["motif_acc"]) < 0.9
    ref = pe.refprob_scores(seqs, g["probs1"], ti)
    np.testing.assert_allclose(ref, g["refprob"], rtol=0, atol=1e-15)
    assert pe.auroc(labels, ref) == pytest.approx(float(g["auroc"]), abs=1e-12)
    assert pe.average_precision(labels, ref) == pytest.approx(float(g["auprc"]), abs=1e-12)
    avg = pe.avg_trueprob_scores(g["probs3"], tt, ml)
    np.testing.assert_allclose(avg, g["avgtrue"], rtol=0, atol=1e-15)
    assert pe.auroc(labels, avg) == pytest.approx(float(g["auroc_avgtrue"]), abs=1e-12)
    sv = pe.sv_llr_boundary(g["left"], g["right"], tab["MutSeq"], g["ref_p"], g["mut_p"], int(g["flanking"]))
    np.testing.assert_allclose(sv, g["sv_scores"], rtol=1e-12, atol=1e-12)
    assert pe.average_precision(labels, sv) == pytest.approx(float(g["sv_auprc"]), abs=1e-12)
    # drivers from saved probability tables (the reference's `logits_path` mode): same numbers, same printed lines
    df = pd.DataFrame({"sequence": seqs, "label": labels})
    p1, p3 = tmp_path / "p1.tsv", tmp_path / "p3.tsv"
    pd.DataFrame(g["probs1"], columns=list("ACGT")).to_csv(p1, sep="\t", index=False)
    pd.DataFrame(g["probs3"], columns=list("ACGT")).to_csv(p3, sep="\t", index=False)
    m = pe.evo_cons(df, token_idx=ti, logits_path=str(p1), metrics_json=str(tmp_path / "m.json"))
    assert m["AUROC"] == pytest.approx(float(g["auroc"]), abs=1e-9) and m["AUPRC"] == pytest.approx(float(g["auprc"]), abs=1e-9)
    assert json.load(open(tmp_path / "m.json"))["token_idx"] == ti
    m = pe.motif_acc(df, mask_idx=pos, motif_len=ml, logits_path=str(p3))
    assert m["motif_accuracy"] == pytest.approx(float(g["motif_acc"]), abs=1e-9)
    m = pe.core_noncore(df, mask_idx=pos, motif_len=ml, logits_path=str(p3), metrics_json=str(tmp_path / "c.json"))
    assert m["AUROC"] == pytest.approx(float(g["auroc_avgtrue"]), abs=1e-9)
    assert m["AUPRC"] == pytest.approx(float(g["auprc_avgtrue"]), abs=1e-9)          # the reference reports both (:523-530)
    assert set(json.load(open(tmp_path / "c.json"))) == {"auroc", "auprc"}
    out = capsys.readouterr().out
    assert "AUROC\t" in out and "token_accuracy\t" in out and "motif_accuracy\t" in out


def test_config1_l20_64_windows_through_cli(tmp_path, monkeypatch):
    """BASELINE config 1 at its stated size: PlantCaduceus_l20 (384-d, 20 layers) CPU forward on 64 synthetic
    512-bp sequences (iid ACGT, default_rng(0)), ref/alt drawn with default_rng(1), mask index 255, synthetic checkpoint seed
    1234, through the zero_shot_score-compatible CLI with -device cpu — plumbing, no GPU.  The model behind the HF surface is
    the C oracle port (the product engine has no CPU path by design); scores are checked against probabilities computed
    outside the CLI."""
    from oracle.c_oracle import COracle, COracleForMaskedLM
    rng = np.random.default_rng(0)
    seqs = ["".join(rng.choice(list("ACGT"), size=512)) for _ in range(64)]
    r1 = np.random.default_rng(1)
    ref = [s[255] for s in seqs]
    alt = [r1.choice([c for c in "ACGT" if c != r]) for r in ref]
    inp, out = tmp_path / "in.tsv", tmp_path / "out.tsv"
    pd.DataFrame({"chr": "chr1", "start": np.arange(64), "end": np.arange(64) + 512, "pos": np.arange(64) + 256, "ref": ref,
                  "alt": alt, "sequences": seqs}).to_csv(inp, sep="\t", index=False)
    cfg = make_config("l20")
    assert (cfg.d_model, cfg.n_layer) == (384, 20)
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    tok = CaduceusTokenizer()
    model = COracleForMaskedLM(sd, cfg, blas=True)
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (model, tok))
    zero_shot.main(["-input-table", str(inp), "-output", str(out), "-model", "PlantCaduceus_l20", "-device", "cpu",
                    "-batchSize", "32"])
    res = pd.read_csv(out, delimiter="\t")
    assert len(res) == 64 and np.isfinite(res["zeroShotScore"]).all()
    k = 16                                                               # rows re-computed outside the CLI
    ids = tok.encode_batch(seqs[:k], mask_index=255)
    lg, _ = COracle(sd, cfg, blas=True).forward(ids)
    z = lg[:, 255, 3:7].astype(np.float64)
    p = np.exp(z - z.max(1, keepdims=True))
    p /= p.sum(1, keepdims=True)
    want = np.log(p[np.arange(k), ["ACGT".index(a) for a in alt[:k]]] / p[np.arange(k), ["ACGT".index(r) for r in ref[:k]]])
    np.testing.assert_allclose(res["zeroShotScore"].to_numpy()[:k], want, rtol=1e-4, atol=1e-5)
    assert np.abs(want).max() > 1e-2                                     # a real signal, not a constant
    # the BLAS-backed and the plain-C GEMM forms of the port agree at full depth
    lg2, _ = COracle(sd, cfg).forward(ids[:2])
    assert np.abs(lg2 - lg[:2]).max() / np.abs(lg[:2]).max() < 1e-4


def test_plantcad2_task_table_sources(tmp_path, monkeypatch):
    """the drivers take a DataFrame, a local table, or the reference's (repo_id, task, split) via HF datasets (:344)"""
    import datasets
    from plantcaduceus_amd import plantcad2_eval as pe
    df = pd.DataFrame({"sequence": ["ACGT" * 4, "TTGA" * 4], "label": [0, 1]})
    p = tmp_path / "t.tsv"
    df.to_csv(p, sep="\t", index=False)
    assert pe._frame(str(p)).equals(df) and pe._frame(df).equals(df)

    class _DS(dict):
        pass
    calls = []

    def fake_load(repo_id, task):
        calls.append((repo_id, task))
        return {"valid": types.SimpleNamespace(to_pandas=lambda: df.iloc[::-1])}
    monkeypatch.setattr(datasets, "load_dataset", fake_load)
    got = pe._frame(("kuleshov-group/cross-species-single-nucleotide-annotation", "conservation", "valid"))
    assert calls == [("kuleshov-group/cross-species-single-nucleotide-annotation", "conservation")]
    assert list(got["label"]) == [1, 0] and list(got.index) == [0, 1]


def test_plantcad2_load_task_through_the_real_datasets_library(tmp_path, monkeypatch):
    """`load_task(repo_id, task, split)` = the reference's `load_dataset(repo_id, task)[split].to_pandas()`
    (src/zero-shot-eval.py:344, :394, :444, :498) executed by the REAL `datasets` package - not a monkey-patch - on a local
    directory laid out as a hub dataset repository is (README.md `configs:` front matter naming one config per task with
    per-split parquet files).  No network: HF_HUB_OFFLINE / HF_DATASETS_OFFLINE; a hub id goes through the same call with the
    user's cache.  Then one driver end to end from that source."""
    datasets = pytest.importorskip("datasets")
    from plantcaduceus_amd import plantcad2_eval as pe
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    monkeypatch.setenv("HF_DATASETS_OFFLINE", "1")
    monkeypatch.setenv("HF_DATASETS_CACHE", str(tmp_path / "cache"))
    repo = tmp_path / "PlantCAD2_zero_shot_tasks"
    (repo / "conservation").mkdir(parents=True)
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), size=40)) for _ in range(10)]
    df = pd.DataFrame({"sequence": seqs, "label": rng.integers(0, 2, size=10)})
    df.to_parquet(repo / "conservation" / "valid.parquet")
    df.iloc[:4].to_parquet(repo / "conservation" / "test.parquet")
    (repo / "README.md").write_text("---\nconfigs:\n- config_name: conservation\n  data_files:\n  - split: valid\n"
                                    "    path: conservation/valid.parquet\n  - split: test\n    path: conservation/test.parquet\n---\n")
    got = pe.load_task(str(repo), "conservation", "valid")
    assert list(got.columns) == ["sequence", "label"] and got["sequence"].tolist() == seqs
    assert len(pe._frame((str(repo), "conservation", "test"))) == 4
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=2), cfg))
    tok = CaduceusTokenizer()
    a = pe.evo_cons((str(repo), "conservation", "valid"), model, tok, "cpu", token_idx=19, batch_size=4)
    b = pe.evo_cons(df, model, tok, "cpu", token_idx=19, batch_size=4)
    assert a == b and set(a) == {"AUROC", "AUPRC"}


def test_plantcad2_command_line(tmp_path, monkeypatch, capsys):
    """`python -m plantcaduceus_amd.plantcad2_eval <sub-command> ...` with the reference's flag names (docs/zero-shot-eval.md):
    same numbers as calling the drivers; `--logits_path` skips the model; the oracle stand-in as the loaded model otherwise."""
    from plantcaduceus_amd import plantcad2_eval as pe
    rng = np.random.default_rng(3)
    n, L = 12, 40
    seqs = ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(n)]
    tab = tmp_path / "task.tsv"
    pd.DataFrame({"sequence": seqs, "label": rng.integers(0, 2, size=n)}).to_csv(tab, sep="\t", index=False)
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=2), cfg))
    tok = CaduceusTokenizer()
    monkeypatch.setattr(pe, "_load_model", lambda m, d: (model, tok))
    want = pe.evo_cons(str(tab), model, tok, "cpu", token_idx=19, batch_size=5)
    got = pe.main(["evo_cons", "--data", str(tab), "--device", "cpu", "--token_idx", "19", "--batch_size", "5",
                   "--save_logits", str(tmp_path / "p.tsv"), "--metrics_json", str(tmp_path / "m.json")])
    assert got == want and set(json.load(open(tmp_path / "m.json"))) == {"auroc", "auprc", "token_idx"}
    monkeypatch.setattr(pe, "_load_model", lambda m, d: (_ for _ in ()).throw(AssertionError("model loaded despite --logits_path")))
    again = pe.main(["evo_cons", "--data", str(tab), "--token-idx", "19", "--logits-path", str(tmp_path / "p.tsv")])
    assert again["AUROC"] == pytest.approx(want["AUROC"], abs=1e-6)
    monkeypatch.setattr(pe, "_load_model", lambda m, d: (model, tok))
    a = pe.main(["motif_acc", "--data", str(tab), "--device", "cpu", "--mask_idx", "18,19,20", "--motif_len", "3"])
    assert a == pe.motif_acc(str(tab), model, tok, "cpu", mask_idx=(18, 19, 20), motif_len=3)
    c = pe.main(["core_noncore", "--data", str(tab), "--device", "cpu", "--mask_idx", "[18,19,20]"])
    assert set(c) == {"AUROC", "AUPRC"}
    with pytest.raises(SystemExit):
        pe.main(["evo_cons", "--device", "cpu"])
    assert "AUROC\t" in capsys.readouterr().out
