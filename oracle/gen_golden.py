#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ — TEST INFRASTRUCTURE, runs ONLY in the build
container (it imports the reference's Python from /root/reference and `transformers`' MambaMixer; neither
travels to the GPU box — only the .npz/.tsv/.json data written here does).

    python oracle/gen_golden.py            # rewrites tests/golden/*

What is pinned, and by what:
  harness_zero_shot.npz   reference `src/zero_shot_score.py` imported with stub modules for the packages the
                          image lacks (`vcf`, `Bio`): its own `zero_shot_score(df, probs)` on
                          `examples/example_snp.tsv` (185 rows after the ref/alt filter of :232) with seeded
                          Dirichlet probabilities, and its own `SequenceDataset` + `extract_logits` loop
                          (:40-62,107-121) driven with the CPU oracle model behind an `encode_plus` shim
                          -> masking index, a/c/g/t column order, softmax-over-4.
  harness_embeddings.npz  reference `src/train_XGBoost.py::extract_embeddings` (:96-114) driven the same way
                          -> hidden_states[-1][:, tokenIdx], fwd/rev channel-reversed averaging.
  harness_windows.json    reference `seq_from_vcf` (:172-214) driven through minimal stand-ins for
                          `vcf.Reader` / `SeqIO` (string slicing semantics) -> window arithmetic and N padding.
  mixer_D*.npz            `transformers.models.mamba.modeling_mamba.MambaMixer` (independent third-party
                          statement of the Mamba-v1 mixer; slow CPU path) with seeded parameters -> in/out.
  harness_plantcad2.npz   reference `src/zero-shot-eval.py` (stub for `fire`): `SingleMaskDataset`/`MultiMaskDataset` +
                          `_masked_probs` and `_unmasked_probs` (:75-178) driven with the CPU oracle model.
  model_tiny.npz          this repo's oracle (literal RCPS form) on a synthetic checkpoint -> logits/hidden;
                          a regression pin for the C oracle and the HIP path (after A==B is tested).
  example_snp.tsv         data file copied from the reference's examples/ (Apache-2.0), config-3 input.
"""
import importlib.util
import json
import os
import shutil
import sys
import types

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import caduceus_oracle as O                                   # noqa: E402
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict  # noqa: E402
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer      # noqa: E402


def import_reference(name, relpath):
    for mod in ("vcf", "Bio", "Bio.SeqIO", "xgboost"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def mixer_param_arrays(D, seed):
    """Seeded parameters of one Mamba mixer (shared with tests/test_oracle.py, which regenerates them)."""
    E, N, R, W = 2 * D, 16, -(-D // 16), 4
    r = np.random.default_rng(seed)
    u = lambda shape, b: r.uniform(-b, b, size=shape).astype(np.float32)
    return dict(
        in_proj=u((2 * E, D), D ** -0.5), conv_w=u((E, 1, W), 0.5), conv_b=u((E,), 0.5),
        x_proj=u((R + 2 * N, E), E ** -0.5), dt_w=u((E, R), R ** -0.5), dt_b=u((E,), 1.0) - 3.0,
        A_log=(np.log(np.tile(np.arange(1, N + 1, dtype=np.float32), (E, 1))) + 0.3 * r.standard_normal((E, N))).astype(np.float32),
        D=r.uniform(0.5, 1.5, E).astype(np.float32), out_proj=u((D, E), E ** -0.5))


def gen_mixer(D, Bsz, L, seed):
    from transformers import MambaConfig
    from transformers.models.mamba.modeling_mamba import MambaMixer
    cfg = MambaConfig(hidden_size=D, state_size=16, conv_kernel=4, expand=2, time_step_rank=-(-D // 16),
                      use_bias=False, use_conv_bias=True, hidden_act="silu", num_hidden_layers=1, vocab_size=8)
    mx = MambaMixer(cfg, layer_idx=0).float().eval()
    p = mixer_param_arrays(D, seed)
    with torch.no_grad():
        mx.in_proj.weight.copy_(torch.from_numpy(p["in_proj"]))
        mx.conv1d.weight.copy_(torch.from_numpy(p["conv_w"]))
        mx.conv1d.bias.copy_(torch.from_numpy(p["conv_b"]))
        mx.x_proj.weight.copy_(torch.from_numpy(p["x_proj"]))
        mx.dt_proj.weight.copy_(torch.from_numpy(p["dt_w"]))
        mx.dt_proj.bias.copy_(torch.from_numpy(p["dt_b"]))
        mx.A_log.copy_(torch.from_numpy(p["A_log"]))
        mx.D.copy_(torch.from_numpy(p["D"]))
        mx.out_proj.weight.copy_(torch.from_numpy(p["out_proj"]))
    x = torch.from_numpy(np.random.default_rng(seed + 1).standard_normal((Bsz, L, D)).astype(np.float32))
    with torch.no_grad():
        y = mx(x)          # eval mode, CPU tensors -> the pure-torch path of MambaMixer.forward
    np.savez_compressed(os.path.join(OUT, f"mixer_D{D}.npz"), x=x.numpy(), y=y.numpy(),
                        D=D, seed=seed, source="transformers %s MambaMixer.forward (pure-torch CPU path)" % __import__("transformers").__version__)
    print(f"mixer_D{D}: y range {float(y.abs().max()):.3f}")


class _TokShim:
    """`encode_plus` surface the reference calls (transformers>=5 removed it)."""
    def __init__(self):
        self.t = CaduceusTokenizer()
        self.mask_token_id = self.t.mask_token_id

    def encode_plus(self, s, return_tensors="pt", **kw):
        return self.t(s, return_tensors=return_tensors)

    def get_vocab(self):
        return self.t.get_vocab()


def gen_harness():
    zs = import_reference("ref_zs", "src/zero_shot_score.py")
    xg = import_reference("ref_xgb", "src/train_XGBoost.py")
    tsv = os.path.join(REF, "examples", "example_snp.tsv")
    shutil.copyfile(tsv, os.path.join(OUT, "example_snp.tsv"))
    df = pd.read_csv(tsv, delimiter="\t")
    keep = df["ref"].isin(list("ACGT")) & df["alt"].isin(list("ACGT"))          # reference :232
    dff = df[keep]
    probs = np.random.default_rng(0).dirichlet(np.ones(4), size=len(dff)).astype(np.float32)
    scores = np.asarray(zs.zero_shot_score(dff, probs), dtype=np.float64)

    # model-in-the-loop: reference Dataset/DataLoader/extract_* driven with the CPU oracle
    cfg_kw = dict(d_model=64, n_layer=2)
    seed = 11
    cfg = make_config("x", **cfg_kw)
    sd = synthetic_state_dict(cfg, seed=seed)
    model = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    tok = _TokShim()
    n_model = 10
    seqs = dff["sequences"].tolist()[:n_model]
    loader = zs.create_dataloader(seqs, tok, 4, 255)
    p_model = zs.extract_logits(model, loader, "cpu", 255, tok)
    s_model = np.asarray(zs.zero_shot_score(dff.iloc[:n_model], p_model), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "harness_zero_shot.npz"), keep_mask=keep.to_numpy(), dirichlet_probs=probs,
                        dirichlet_scores=scores, model_d_model=cfg_kw["d_model"], model_n_layer=cfg_kw["n_layer"],
                        model_seed=seed, model_rows=n_model, model_probs=p_model.astype(np.float32),
                        model_scores=s_model, token_idx=255)
    loader2 = xg.create_dataloader(seqs, tok, 4)
    emb = xg.extract_embeddings(model, loader2, "cpu", 255)
    np.savez_compressed(os.path.join(OUT, "harness_embeddings.npz"), model_d_model=cfg_kw["d_model"],
                        model_n_layer=cfg_kw["n_layer"], model_seed=seed, model_rows=n_model, token_idx=255,
                        embeddings=np.ascontiguousarray(emb, dtype=np.float32))
    print("harness: kept", int(keep.sum()), "rows; model probs[0] =", p_model[0])

    # window extraction: reference seq_from_vcf through stand-ins for vcf.Reader / SeqIO
    genome = {"chr1": "".join(np.random.default_rng(5).choice(list("ACGTacgtN"), size=1500)),
              "chr2": "".join(np.random.default_rng(6).choice(list("ACGT"), size=700))}
    recs = [("chr1", 1, "A", ["C"]), ("chr1", 100, "C", ["G", "T"]), ("chr1", 256, "G", ["A"]), ("chr1", 257, "G", ["A"]),
            ("chr1", 800, "T", ["TA"]), ("chr1", 1300, "A", ["G", "AT"]), ("chr1", 1500, "C", ["T"]),
            ("chr2", 300, "A", ["C"]), ("chr2", 699, "G", ["<DEL>"]), ("chr2", 700, "T", ["C"])]

    class _Alt:
        def __init__(self, s):
            self.sequence = s
            self.type = "SNV" if len(s) == 1 and s in "ACGT" else "INDEL"

    class _Rec:
        def __init__(self, c, p, r, alts):
            self.CHROM, self.POS, self.REF, self.ALT = c, p, r, [_Alt(a) for a in alts]

    class _SeqRec:   # BioPython SeqRecord slicing semantics: python slice of the underlying string
        def __init__(self, s):
            self.seq = s

        def __getitem__(self, sl):
            return _SeqRec(self.seq[sl])

    sys.modules["vcf"].Reader = lambda filename=None: iter([_Rec(*r) for r in recs])
    sys.modules["Bio.SeqIO"].parse = lambda f, fmt: None
    sys.modules["Bio.SeqIO"].to_dict = lambda it: {k: _SeqRec(v) for k, v in genome.items()}
    zs.vcf = sys.modules["vcf"]
    zs.SeqIO = sys.modules["Bio.SeqIO"]
    windows = {}
    for tidx in (255, 100):
        args = types.SimpleNamespace(inputVCF="x.vcf", inputFasta="x.fa", tokenIdx=tidx)
        seqs_w, ridx = zs.seq_from_vcf(args)
        windows[str(tidx)] = dict(sequences=seqs_w, record_indices=ridx)
    with open(os.path.join(OUT, "harness_windows.json"), "w") as f:
        json.dump(dict(genome=genome, records=recs, windows=windows), f)
    print("windows:", {k: len(v["sequences"]) for k, v in windows.items()})


def gen_plantcad2():
    """reference src/zero-shot-eval.py: Single/MultiMaskDataset + _masked_probs and _unmasked_probs, oracle model in the loop"""
    sys.modules.setdefault("fire", types.ModuleType("fire"))
    ev = import_reference("ref_eval", "src/zero-shot-eval.py")
    from torch.utils.data import DataLoader
    df = pd.read_csv(os.path.join(REF, "examples", "example_snp.tsv"), delimiter="\t")
    seqs = df["sequences"].iloc[:6]
    cfg_kw = dict(d_model=64, n_layer=2)
    seed = 11
    cfg = make_config("x", **cfg_kw)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=seed), cfg))
    tok = CaduceusTokenizer()
    single = ev._masked_probs(model, tok, DataLoader(ev.SingleMaskDataset(seqs, tok, 255), batch_size=4), "cpu")
    multi_idx = [300, 10, 255]
    multi = ev._masked_probs(model, tok, DataLoader(ev.MultiMaskDataset(seqs, tok, multi_idx), batch_size=4), "cpu")
    unm = ev._unmasked_probs(seqs, tok, model, "cpu", 4)
    np.savez_compressed(os.path.join(OUT, "harness_plantcad2.npz"), model_d_model=cfg_kw["d_model"],
                        model_n_layer=cfg_kw["n_layer"], model_seed=seed, rows=6, single_idx=255, single=single.astype(np.float32),
                        multi_idx=np.array(multi_idx), multi=multi.astype(np.float32), unmasked=unm.astype(np.float32))
    print("plantcad2:", single.shape, multi.shape, unm.shape)


def gen_plantcad2_metrics():
    """reference src/zero-shot-eval.py:181-320: the task metrics (token / motif accuracy, AUROC, AUPRC, ref-prob and
    true-prob scores, SV boundary LLR) on seeded synthetic tables -> inputs (json) + the reference's outputs (npz)."""
    import json
    sys.modules.setdefault("fire", types.ModuleType("fire"))
    ev = import_reference("ref_eval_m", "src/zero-shot-eval.py")
    from sklearn.metrics import average_precision_score
    rng = np.random.default_rng(21)
    n, Ls, positions, motif_len = 48, 40, [17, 18, 19], 3
    alphabet = np.array(list("ACGTacgtNn"))
    seqs = ["".join(rng.choice(alphabet, size=Ls, p=[.2, .2, .2, .2, .04, .04, .04, .04, .02, .02])) for _ in range(n)]
    labels = rng.integers(0, 2, size=n)
    probs3 = rng.dirichlet([0.7] * 4, size=n * motif_len)
    probs1 = np.round(rng.dirichlet([0.7] * 4, size=n), 2)          # rounding -> tied scores
    probs1 /= probs1.sum(1, keepdims=True)
    df = pd.DataFrame({"sequence": seqs, "label": labels})
    tt = ev._compute_true_tokens_from_seq(df["sequence"], positions)
    for r, b in enumerate(tt):                                      # a model that is right about 70 % of the time
        if b in "ACGT" and rng.random() < 0.7:
            probs3[r, "ACGT".index(b)] += 2.0
    probs3 /= probs3.sum(1, keepdims=True)
    out = dict(token_acc=ev._metric_token_accuracy(probs3, tt), motif_acc=ev._metric_motif_accuracy(probs3, tt, motif_len),
               auroc=ev._compute_auroc(df, probs1, 18, "sequence"), refprob=ev._refprob_scores(df, probs1, 18, "sequence"),
               avgtrue=ev._avg_trueprob_scores(probs3, tt, motif_len))
    out["auprc"] = float(average_precision_score(labels, out["refprob"]))
    out["auroc_avgtrue"] = float(ev.auc(*ev.roc_curve(labels, out["avgtrue"])[:2]))
    out["auprc_avgtrue"] = float(average_precision_score(labels, out["avgtrue"]))   # core_noncore prints both (:523-530)
    # SV: windows of length Lw, central 2*flanking positions mutated; left/right 1-based boundaries with room for the flanks
    Lw, F = 48, 5
    ref_seqs = ["".join(rng.choice(alphabet, size=Lw, p=[.2, .2, .2, .2, .04, .04, .04, .04, .02, .02])) for _ in range(n)]
    mut_seqs = ["".join(rng.choice(alphabet, size=Lw, p=[.2, .2, .2, .2, .04, .04, .04, .04, .02, .02])) for _ in range(n)]
    left = rng.integers(F + 1, 18, size=n)
    right = rng.integers(24, Lw - F, size=n)
    ref_p = rng.dirichlet([0.5] * 4, size=(n, Lw))
    mut_p = rng.dirichlet([0.5] * 4, size=(n, Lw))
    mut_p[0, Lw // 2, :] = [0.0, 1.0, 0.0, 0.0]                    # exercises the 1e-12 floor
    sv_df = pd.DataFrame({"RefSeq": ref_seqs, "MutSeq": mut_seqs, "left": left, "right": right, "label": labels})
    out["sv_scores"] = ev._sv_llr_boundary(sv_df, ref_p, mut_p, F)
    out["sv_auprc"] = float(average_precision_score(labels, out["sv_scores"]))
    np.savez_compressed(os.path.join(OUT, "harness_plantcad2_metrics.npz"), labels=labels, probs3=probs3, probs1=probs1,
                        positions=np.array(positions), motif_len=motif_len, token_idx=18, left=left, right=right, ref_p=ref_p,
                        mut_p=mut_p, flanking=F, true_tokens=tt, **{k: np.asarray(v) for k, v in out.items()})
    with open(os.path.join(OUT, "harness_plantcad2_metrics.json"), "w") as f:
        json.dump({"sequences": seqs, "RefSeq": ref_seqs, "MutSeq": mut_seqs}, f)
    print("plantcad2 metrics:", {k: (float(v) if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


def gen_model_tiny():
    cfg = make_config("x", d_model=64, n_layer=3)
    sd = synthetic_state_dict(cfg, seed=2024)
    ids = torch.from_numpy(np.random.default_rng(9).integers(1, 7, size=(3, 24)).astype(np.int64))
    ids[0, 5] = 2
    out = O.forward_literal(ids, O.params_from_state_dict(sd, cfg), output_hidden_states=True)
    np.savez_compressed(os.path.join(OUT, "model_tiny.npz"), d_model=64, n_layer=3, seed=2024, ids=ids.numpy(),
                        logits=out["logits"].numpy(), hidden=out["hidden"].numpy(),
                        hidden0=out["all_hidden"][0].numpy(), hidden1=out["all_hidden"][1].numpy())
    print("model_tiny: logits range", float(out["logits"].abs().max()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    gen_mixer(32, 2, 24, 100)
    gen_mixer(384, 2, 48, 200)
    gen_model_tiny()
    gen_harness()
    gen_plantcad2()
    gen_plantcad2_metrics()
    print("golden vectors written to", OUT)
