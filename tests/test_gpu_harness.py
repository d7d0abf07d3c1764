"""-m gpu: the committed golden vectors (reference-harness outputs, model_tiny) and BASELINE config 3
(l32 bf16 on examples/example_snp.tsv) through the HIP path behind the HF surface."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from plantcaduceus_amd import embeddings, zero_shot
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_model(cfg, sd, dtype):
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    return m.to(dtype).to(DEV)


def test_model_tiny_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    cfg = make_config("x", d_model=int(g["d_model"]), n_layer=int(g["n_layer"]))
    cfg.materialize_all_hidden_states = True
    m = hip_model(cfg, synthetic_state_dict(cfg, seed=int(g["seed"])), torch.float32)
    out = m(input_ids=torch.from_numpy(g["ids"]).to(DEV), output_hidden_states=True)
    lg = out.logits.cpu().numpy()
    assert np.abs(lg - g["logits"]).max() / np.abs(g["logits"]).max() < 1e-4          # north_star: <=1e-4 rel
    assert (lg[..., 3:7].argmax(-1) == g["logits"][..., 3:7].argmax(-1)).all()        # bit-exact token calls
    hs = out.hidden_states
    assert len(hs) == cfg.n_layer + 1
    for got, key in ((hs[-1], "hidden"), (hs[0], "hidden0"), (hs[1], "hidden1")):
        assert np.abs(got.cpu().numpy() - g[key]).max() / np.abs(g[key]).max() < 1e-4


def test_reference_harness_goldens_through_hip(golden_dir):
    z = np.load(os.path.join(golden_dir, "harness_zero_shot.npz"))
    e = np.load(os.path.join(golden_dir, "harness_embeddings.npz"))
    df = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")
    df = df[z["keep_mask"]].iloc[: int(z["model_rows"])]
    cfg = make_config("x", d_model=int(z["model_d_model"]), n_layer=int(z["model_n_layer"]))
    m = hip_model(cfg, synthetic_state_dict(cfg, seed=int(z["model_seed"])), torch.float32)
    tok = CaduceusTokenizer()
    probs = zero_shot.extract_logits(m, df["sequences"].tolist(), DEV, int(z["token_idx"]), tok, batch_size=4)
    np.testing.assert_allclose(probs, z["model_probs"], rtol=1e-4, atol=1e-6)
    assert (probs.argmax(1) == z["model_probs"].argmax(1)).all()
    np.testing.assert_allclose(np.asarray(zero_shot.zero_shot_score(df, probs)), z["model_scores"], rtol=1e-3, atol=1e-4)
    emb = embeddings.extract_embeddings(m, df["sequences"].tolist(), DEV, int(e["token_idx"]), tok, batch_size=4)
    assert np.abs(emb - e["embeddings"]).max() / np.abs(e["embeddings"]).max() < 1e-4


def test_config3_l32_bf16_example_snps(golden_dir):
    """PlantCaduceus_l32 bf16 on the reference's example table (185 scored rows), synthetic checkpoint seed 1234.
    Size-independent properties on all rows + the fp32 C oracle on a bounded sample."""
    df = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")
    df = df[df["ref"].isin(list("ACGT")) & df["alt"].isin(list("ACGT"))]
    assert len(df) == 185
    cfg = make_config("l32")
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    m = hip_model(cfg, sd, torch.bfloat16)
    tok = CaduceusTokenizer()
    seqs = df["sequences"].tolist()
    probs = zero_shot.extract_logits(m, seqs, DEV, 255, tok, batch_size=128)
    assert probs.shape == (185, 4) and np.isfinite(probs).all()
    np.testing.assert_allclose(probs.sum(1), 1.0, rtol=1e-5)
    scores = np.asarray(zero_shot.zero_shot_score(df, probs))
    assert np.isfinite(scores).all()
    # reverse-complement equivariance of the call: P(rc window masked at 511-255)[comp base] == P(window)[base]
    comp = str.maketrans("ACGT", "TGCA")
    rc = [s.translate(comp)[::-1] for s in seqs]
    probs_rc = zero_shot.extract_logits(m, rc, DEV, 511 - 255, tok, batch_size=128)[:, ::-1]
    assert np.abs(probs_rc - probs).max() < 2e-2           # bf16 path: both strands are computed, in swapped roles
    # bounded sample against the CPU oracle (C port) emulating the bf16 model: bf16 weights, bf16 rounding at the
    # reference's tensor boundaries, fp32 arithmetic in between; 32 layers deep.  Tolerance on probabilities: a few
    # bf16 ulps of accumulated rounding-order noise; argmax where the oracle's margin exceeds it.
    from oracle.c_oracle import COracle
    n = 16
    ids = tok.encode_batch(seqs[:n], mask_index=255)
    lg, _ = COracle(sd, cfg, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True, blas=True).forward(ids)
    z = lg[:, 255, 3:7]
    ref = np.exp(z - z.max(1, keepdims=True))
    ref /= ref.sum(1, keepdims=True)
    print("config3: max |p_hip - p_oracle(bf16-emulating)| =", np.abs(probs[:n] - ref).max())
    assert np.abs(probs[:n] - ref).max() < 1e-2
    top2 = np.sort(ref, 1)[:, -2:]
    conf = (top2[:, 1] - top2[:, 0]) > 2e-2          # twice the probability tolerance
    assert conf.sum() >= n // 2, "vacuous argmax check"
    assert (probs[:n].argmax(1)[conf] == ref.argmax(1)[conf]).all()
    print("argmax agreement on all %d sampled rows: %.3f" % (n, (probs[:n].argmax(1) == ref.argmax(1)).mean()))


def test_config2_l20_bf16_batch1024():
    """BASELINE config 2: PlantCaduceus_l20 bf16, 1024 synthetic 512-bp windows (numpy default_rng(0)), mask at 255.
    All rows: finite, normalised, reverse-complement equivariant.  A bounded sample against the CPU oracle port in its
    bf16-emulating mode (bf16 weights and tensor boundaries, fp32 arithmetic in between)."""
    from oracle.c_oracle import COracle
    cfg = make_config("l20")
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    m = hip_model(cfg, sd, torch.bfloat16)
    tok = CaduceusTokenizer()
    rng = np.random.default_rng(0)
    ids = rng.integers(3, 7, size=(1024, 512)).astype(np.int32)
    ids[:, 255] = tok.mask_token_id
    probs = zero_shot.extract_logits(m, ids, DEV, 255, tok, batch_size=1024)
    assert probs.shape == (1024, 4) and np.isfinite(probs).all()
    np.testing.assert_allclose(probs.sum(1), 1.0, rtol=1e-5)
    comp = np.array(cfg.complement_list(), dtype=np.int32)
    rc = comp[ids[:, ::-1]]
    probs_rc = zero_shot.extract_logits(m, np.ascontiguousarray(rc), DEV, 511 - 255, tok, batch_size=1024)[:, ::-1]
    assert np.abs(probs_rc - probs).max() < 2e-2
    n = 32
    lg, _ = COracle(sd, cfg, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True, blas=True).forward(ids[:n])
    z = lg[:, 255, 3:7]
    ref = np.exp(z - z.max(1, keepdims=True))
    ref /= ref.sum(1, keepdims=True)
    print("config2: max |p_hip - p_oracle(bf16-emulating)| =", np.abs(probs[:n] - ref).max())
    assert np.abs(probs[:n] - ref).max() < 1e-2
    top2 = np.sort(ref, 1)[:, -2:]
    conf = (top2[:, 1] - top2[:, 0]) > 2e-2          # twice the probability tolerance
    assert conf.sum() >= n // 2, "vacuous argmax check"
    assert (probs[:n].argmax(1)[conf] == ref.argmax(1)[conf]).all()
    print("argmax agreement on all %d sampled rows: %.3f" % (n, (probs[:n].argmax(1) == ref.argmax(1)).mean()))


def test_plantcad2_helpers_through_hip(golden_dir):
    from plantcaduceus_amd import plantcad2_eval as pe
    g = np.load(os.path.join(golden_dir, "harness_plantcad2.npz"))
    df = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")
    seqs = df["sequences"].tolist()[: int(g["rows"])]
    cfg = make_config("x", d_model=int(g["model_d_model"]), n_layer=int(g["model_n_layer"]))
    m = hip_model(cfg, synthetic_state_dict(cfg, seed=int(g["model_seed"])), torch.float32)
    tok = CaduceusTokenizer()
    np.testing.assert_allclose(pe.masked_probs(m, tok, seqs, g["multi_idx"].tolist(), DEV, batch_size=4), g["multi"], rtol=1e-4, atol=1e-6)
    un = pe.unmasked_probs(seqs, tok, m, DEV, batch_size=4)
    np.testing.assert_allclose(un, g["unmasked"], rtol=1e-4, atol=1e-6)
    assert (un.argmax(-1) == g["unmasked"].argmax(-1)).all()


def test_plantcad2_sv_effect_and_motif_tasks_through_hip(golden_dir, tmp_path):
    """task drivers end to end on the HIP model: sv_effect (un-masked probabilities of RefSeq / MutSeq -> boundary LLR ->
    AUPRC) and motif_acc (multi-mask) agree with the same drivers' arithmetic on the oracle model's probabilities."""
    from plantcaduceus_amd import plantcad2_eval as pe
    import oracle.caduceus_oracle as O
    cfg = make_config("x", d_model=64, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=5)
    m = hip_model(cfg, sd, torch.float32)
    om = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    tok = CaduceusTokenizer()
    rng = np.random.default_rng(4)
    n, L, F = 10, 64, 5
    mk = lambda: "".join(rng.choice(list("ACGTN"), size=L, p=[.24, .24, .24, .24, .04]))
    df = pd.DataFrame({"RefSeq": [mk() for _ in range(n)], "MutSeq": [mk() for _ in range(n)],
                       "left": rng.integers(F + 1, 24, size=n), "right": rng.integers(36, L - F, size=n),
                       "label": rng.integers(0, 2, size=n)})
    out = tmp_path / "sv.tsv"
    res = pe.sv_effect(df, m, tok, DEV, batch_size=4, flanking=F, output=str(out))
    ref_scores = pe.sv_llr_boundary(df["left"], df["right"], df["MutSeq"], pe.unmasked_probs(df["RefSeq"], tok, om, "cpu", 4),
                                    pe.unmasked_probs(df["MutSeq"], tok, om, "cpu", 4), F)
    got = pd.read_csv(out, sep="\t")["score"].to_numpy()
    np.testing.assert_allclose(got, ref_scores, rtol=1e-3, atol=1e-5)
    assert res["AUPRC"] == pytest.approx(pe.average_precision(df["label"], ref_scores), abs=1e-6)
    d2 = pd.DataFrame({"sequence": df["RefSeq"], "label": df["label"]})
    a = pe.motif_acc(d2, m, tok, DEV, mask_idx=(30, 31, 32), motif_len=3, batch_size=4)
    b = pe.motif_acc(d2, om, tok, "cpu", mask_idx=(30, 31, 32), motif_len=3, batch_size=4)
    assert a == b


def test_cli_end_to_end_from_snapshot_on_gpu(golden_dir, tmp_path):
    """the reference's command line on the GPU, nothing mocked: a snapshot directory (config.json + model.safetensors under
    the reference's key names) -> `zero_shot.main` (-input-table, -device cuda:0) -> TSV; scores against the oracle run on the
    same checkpoint.  Also the embedding script's extract step on the model `load_model_and_tokenizer` returns."""
    import oracle.caduceus_oracle as O
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    d = str(tmp_path / "snap")
    cfg, sd = make_synthetic_checkpoint(d, "x", seed=13, stress=False, d_model=128, n_layer=2)
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:24]
    inp, out = tmp_path / "in.tsv", tmp_path / "out.tsv"
    src.to_csv(inp, sep="\t", index=False)
    zero_shot.main(["-input-table", str(inp), "-output", str(out), "-model", d, "-device", DEV, "-batchSize", "7"])
    res = pd.read_csv(out, delimiter="\t")
    ok = (src["ref"].isin(list("ACGT")) & src["alt"].isin(list("ACGT"))).to_numpy()
    assert len(res) == int(ok.sum())
    om = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    tok = CaduceusTokenizer()
    ids = torch.from_numpy(zero_shot.tokenize_masked(src["sequences"].tolist(), tok, 255).astype(np.int64))
    p = torch.softmax(om(input_ids=ids).logits[:, 255, 3:7].float(), -1).numpy()
    want = np.array([np.log(p[i, "ACGT".index(a)] / p[i, "ACGT".index(r)])
                     for i, (r, a) in enumerate(zip(src["ref"], src["alt"])) if ok[i]])
    # the dtype policy picks bf16 on this GPU (get_optimal_dtype): bf16 tolerance on a log-ratio of probabilities
    np.testing.assert_allclose(res["zeroShotScore"].to_numpy(), want, rtol=0, atol=5e-2)
    model, tok2 = zero_shot.load_model_and_tokenizer(d, DEV)
    emb = embeddings.extract_embeddings(model, src["sequences"].tolist()[:5], DEV, 255, tokenizer=tok2, batch_size=4)
    assert emb.shape == (5, cfg.d_model) and np.isfinite(emb).all()
