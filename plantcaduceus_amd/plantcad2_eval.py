"""Probability primitives of the reference's PlantCAD2 evaluation CLI (`src/zero-shot-eval.py`), on the same
`model(input_ids=...).logits` boundary as the zero-shot path:

  masked_probs    `SingleMaskDataset` / `MultiMaskDataset` + `_masked_probs`  (:75-140): mask one or several positions of
                  every sequence, softmax over a,c,g,t of the logits AT the masked positions -> [N * n_mask, 4], rows in
                  (sequence, ascending position) order (the order of the reference's `torch.masked_select`)
  unmasked_probs  `_unmasked_probs` (:143-178): per-position probabilities of the un-masked sequence -> [N, L, 4]

Only the head rows that are read are evaluated (`positions=`); tokenisation is the vectorised LUT.

Task metrics and drivers on top of them (same file, `:181-320` and `ZeroShotEval` `:323-530`), vectorised numpy, no
sklearn: `true_tokens`, `token_accuracy`, `motif_accuracy`, `refprob_scores`, `auroc`, `average_precision`,
`avg_trueprob_scores`, `sv_llr_boundary`, and the four tasks `evo_cons`, `motif_acc`, `core_noncore`, `sv_effect`, which
take a DataFrame, a local .tsv/.csv/.parquet path, or the reference's `(repo_id, task[, split])` (HF `datasets`: cache or network) and print / return
the reference's metric lines.  Pinned by `tests/golden/harness_plantcad2_metrics.*` (outputs of the reference's functions).
"""
from __future__ import annotations

import json
import logging
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import sharding
from .zero_shot import check_model_inputs, tokenize_masked

NUCLEOTIDES_LOWER = ("a", "c", "g", "t")


GATHER_CHUNK = 4096      # windows per rank between two all-gathers of the sharded loops (not per batch: one collective per chunk)


def _sharded_rows(n_total: int, run_rows, out: np.ndarray, chunk: int = GATHER_CHUNK) -> np.ndarray:
    """The reference's loops (`_masked_probs` / `_unmasked_probs`, src/zero-shot-eval.py:129-178) are single-device loops over
    independent windows.  Here every rank evaluates its contiguous block of the `n_total` windows (sharding.shard_bounds, as the
    zero-shot path does: SURVEY.md §8(e)) in chunks of at most `chunk` windows; after each chunk ONE all-gather hands every rank
    every rank's rows of that chunk, which are copied to their single-process position in the host array `out` - so every rank
    returns the rows in the single-GPU order and the row arithmetic is the single-GPU one (windows are independent).
    run_rows(lo, hi) -> device tensor [hi - lo, ...] for the global rows [lo, hi) (hi == lo: an empty tensor of the right trailing
    shape).  Every rank runs the same number of gathers (chunks are cut over the common block length; short tails are padded with
    a dummy row that is stripped).  Outside a process group: a plain loop."""
    rank, ws = sharding.world()
    start, stop, per = sharding.shard_bounds(n_total, rank, ws)
    grouped = ws > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized())
    for c0 in range(0, per, max(1, chunk)):
        c1 = min(c0 + chunk, per)
        lo, hi = min(start + c0, stop), min(start + c1, stop)
        local = run_rows(lo, hi)
        if not grouped:
            out[lo:hi] = local.cpu().numpy()
            continue
        g = sharding.all_gather_blocks(sharding.pad_rows(local, c1 - c0)).cpu().numpy()
        for r in range(ws):
            a, b = r * per + c0, min(r * per + c1, n_total)
            if b > a:
                out[a:b] = g[r * (c1 - c0): r * (c1 - c0) + (b - a)]
    return out


def masked_probs(model, tokenizer, sequences: Sequence[str], mask_idx: Union[int, Sequence[int]], device,
                 batch_size: int = 128) -> np.ndarray:
    """Sharded under torchrun: each rank tokenises and evaluates its own block of the sequences only; all ranks return all rows."""
    idx = sorted({int(mask_idx)} if isinstance(mask_idx, (int, np.integer)) else {int(i) for i in mask_idx})
    cols = [tokenizer.get_vocab()[n] for n in NUCLEOTIDES_LOWER]
    seqs = list(sequences)
    n = len(seqs)
    fast = bool(getattr(model, "supports_positions", False)) and len(idx) <= 16

    def run_rows(lo, hi):
        if hi <= lo:
            return torch.zeros((0, len(idx), 4), dtype=torch.float32, device=device)
        ids = torch.from_numpy(tokenize_masked(seqs[lo:hi], tokenizer, None).astype(np.int64))
        if ids.shape[1] <= max(idx):
            raise ValueError(f"mask index {max(idx)} out of range for sequence length {ids.shape[1]}")
        ids[:, idx] = tokenizer.mask_token_id
        parts = []
        for b0 in range(0, ids.shape[0], batch_size):
            cur = ids[b0:b0 + batch_size].to(device)
            lg = model(input_ids=cur, positions=idx).logits if fast else model(input_ids=cur).logits[:, idx, :]
            parts.append(torch.softmax(lg[..., cols].float(), dim=-1))
        return torch.cat(parts, dim=0)

    if n and min(len(str(q)) for q in seqs) <= max(idx):          # the reference's index error, raised on EVERY rank before any collective
        raise ValueError(f"mask index {max(idx)} out of range for sequence length {min(len(str(q)) for q in seqs)}")
    out = np.zeros((n, len(idx), 4), dtype=np.float32)
    with torch.inference_mode():
        _sharded_rows(n, run_rows, out)
    check_model_inputs(model)   # the engine validates token ids on the device; raise here, where results are read back (bits OR-ed over the ranks)
    return out.reshape(-1, 4)


def unmasked_probs(sequences: Sequence[str], tokenizer, model, device, batch_size: int = 32) -> np.ndarray:
    """Sharded under torchrun like masked_probs; [n, L, 4] probabilities are gathered per chunk of <= GATHER_CHUNK windows per rank."""
    cols = [tokenizer.get_vocab()[n] for n in NUCLEOTIDES_LOWER]
    seqs = [str(s) for s in sequences]
    if len({len(s) for s in seqs}) > 1:
        raise ValueError("All sequences must have same length")
    L = len(seqs[0]) if seqs else 0

    def run_rows(lo, hi):
        if hi <= lo:
            return torch.zeros((0, L, 4), dtype=torch.float32, device=device)
        ids = torch.from_numpy(tokenize_masked(seqs[lo:hi], tokenizer, None).astype(np.int64))
        parts = []
        for b0 in range(0, ids.shape[0], batch_size):
            lg = model(input_ids=ids[b0:b0 + batch_size].to(device)).logits[..., cols]
            parts.append(torch.softmax(lg.float(), dim=-1))
        return torch.cat(parts, dim=0)

    out = np.zeros((len(seqs), L, 4), dtype=np.float32)
    # [rows, L, 4] fp32 per rank and chunk on the device: bound the chunk by bytes as well (8 192-bp windows: 128 KiB per window)
    chunk = max(1, min(GATHER_CHUNK, (1 << 30) // max(1, L * 16)))
    with torch.inference_mode():
        _sharded_rows(len(seqs), run_rows, out, chunk)
    check_model_inputs(model)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# task metrics (reference src/zero-shot-eval.py:181-320)
NUCLEOTIDES = ("A", "C", "G", "T")
_CODE = np.full(256, -1, dtype=np.int64)
for _i, _b in enumerate(NUCLEOTIDES):
    _CODE[ord(_b)] = _i
    _CODE[ord(_b.lower())] = _i


def _codes(sequences, positions) -> np.ndarray:
    """[N, len(positions)] index into A,C,G,T of sequence[p] (case-insensitive), -1 for anything else."""
    pos = np.asarray(list(positions), dtype=np.int64)
    seqs = [str(x) for x in sequences]
    out = np.full((len(seqs), len(pos)), -1, dtype=np.int64)
    for i, q in enumerate(seqs):                       # strings may differ in length; positions must exist (as in the reference)
        b = np.frombuffer(q.encode("latin-1", "replace"), dtype=np.uint8)
        out[i] = _CODE[b[pos]]
    return out


def true_tokens(sequences, positions: Sequence[int]) -> np.ndarray:
    """`_compute_true_tokens_from_seq` (:246-251): upper-cased base at every position of every sequence, row-major."""
    return np.array([str(q)[int(p)].upper() for q in sequences for p in positions])


def _token_codes(tokens) -> np.ndarray:
    t = np.asarray(tokens)
    lut = {b: i for i, b in enumerate(NUCLEOTIDES)}
    return np.array([lut.get(str(x), -1) for x in t], dtype=np.int64)


def token_accuracy(probs: np.ndarray, tokens) -> float:
    """`_metric_token_accuracy` (:254-261): arg-max call vs truth over the positions whose truth is A/C/G/T."""
    c = _token_codes(tokens)
    ok = c >= 0
    if not ok.any():
        return 0.0
    return float((np.asarray(probs).argmax(axis=1)[ok] == c[ok]).mean())


def motif_accuracy(probs: np.ndarray, tokens, motif_len: int) -> float:
    """`_metric_motif_accuracy` (:264-274): all positions of a motif called right, over motifs without unknown bases."""
    c = _token_codes(tokens)
    if len(c) % motif_len:
        raise AssertionError("total masked positions not divisible by motif_len")
    c = c.reshape(-1, motif_len)
    pred = np.asarray(probs).argmax(axis=1).reshape(-1, motif_len)
    ok = (c >= 0).all(axis=1)
    if not ok.any():
        return 0.0
    return float((pred[ok] == c[ok]).all(axis=1).mean())


def refprob_scores(sequences, probs: np.ndarray, token_idx: int) -> np.ndarray:
    """`_refprob_scores` (:290-298): probability of the sequence's own base at the masked index (0 for N etc.)."""
    c = _codes(sequences, [token_idx])[:, 0]
    p = np.asarray(probs).reshape(len(c), -1)
    out = np.zeros(len(c), dtype=float)
    ok = c >= 0
    out[ok] = p[ok, c[ok]]
    return out


def _midranks(x: np.ndarray) -> np.ndarray:
    order = np.argsort(x, kind="mergesort")
    xs = x[order]
    first = np.r_[True, xs[1:] != xs[:-1]]
    start = np.flatnonzero(first)
    end = np.r_[start[1:], len(xs)]
    r = np.empty(len(xs), dtype=float)
    for a, b in zip(start, end):
        r[a:b] = 0.5 * (a + b - 1) + 1.0
    out = np.empty(len(xs), dtype=float)
    out[order] = r
    return out


def auroc(y_true, scores) -> float:
    """area under `roc_curve` (`_compute_auroc` :277-288) = Mann-Whitney statistic with mid-ranks for ties."""
    y = np.asarray(y_true).astype(int)
    s = np.asarray(scores, dtype=float)
    n1, n0 = int((y == 1).sum()), int((y == 0).sum())
    if n1 == 0 or n0 == 0:
        return float("nan")
    r = _midranks(s)
    return float((r[y == 1].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def average_precision(y_true, scores) -> float:
    """sklearn's `average_precision_score`: sum over distinct thresholds of (recall step) x precision."""
    y = np.asarray(y_true).astype(int)
    s = np.asarray(scores, dtype=float)
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    last = np.r_[s[1:] != s[:-1], True]                 # last element of every tie group
    tp = np.cumsum(y)[last].astype(float)
    n = (np.flatnonzero(last) + 1).astype(float)
    npos = float(y.sum())
    if npos == 0:
        return 0.0
    recall = tp / npos
    return float(np.sum(np.diff(np.r_[0.0, recall]) * (tp / n)))


def avg_trueprob_scores(probs: np.ndarray, tokens, motif_len: int) -> np.ndarray:
    """`_avg_trueprob_scores` (:301-320): mean probability of the true base over the masked positions of each example."""
    c = _token_codes(tokens)
    if len(c) % motif_len:
        raise AssertionError("total masked positions not divisible by motif_len")
    p = np.asarray(probs)
    v = np.zeros(len(c), dtype=float)
    ok = c >= 0
    v[ok] = p[np.flatnonzero(ok), c[ok]]
    return v.reshape(-1, motif_len).mean(axis=1)


def sv_llr_boundary(left, right, mut_seqs, ref_probs: np.ndarray, mut_probs: np.ndarray, flanking: int) -> np.ndarray:
    """`_sv_llr_boundary` (:181-243): -mean over 2*flanking positions of log(p_mut / p_ref) of the MUTATED sequence's base;
    ref windows (1-based `left`, `right`): [left-flanking, left-1] and [right+1, right+flanking]; mut window: the central
    2*flanking positions of the mutated sequence; unknown bases contribute 0; probabilities floored at 1e-12."""
    ref_probs, mut_probs = np.asarray(ref_probs), np.asarray(mut_probs)
    n, L = ref_probs.shape[0], ref_probs.shape[1]
    c0 = L // 2
    k = np.arange(flanking)
    left = np.asarray(left, dtype=np.int64)
    right = np.asarray(right, dtype=np.int64)
    ref_pos = np.concatenate([left[:, None] - flanking + k[None, :] - 1, right[:, None] + k[None, :]], axis=1)   # 0-based
    mut_pos = np.arange(c0 - flanking, c0 + flanking)
    base = _codes(mut_seqs, mut_pos)                                                # [n, 2F]
    rows = np.arange(n)[:, None]
    j = np.where(base >= 0, base, 0)
    r = np.maximum(ref_probs[rows, ref_pos, j], 1e-12)
    m = np.maximum(mut_probs[rows, mut_pos[None, :], j], 1e-12)
    llr = np.where(base >= 0, np.log(m / r), 0.0)
    return -llr.mean(axis=1)


# ---------------------------------------------------------------------------------------------------------------------
# task drivers (reference `ZeroShotEval` :323-530); `data`: DataFrame or path to a local .tsv / .csv / .parquet table
def load_task(repo_id: str, task: str, split: str = "valid"):
    """`load_dataset(repo_id, task)[split].to_pandas()` — how every sub-command of the reference gets its table (:344, :394,
    :444, :498).  Needs the HF `datasets` cache or a network; the drivers below also take a DataFrame or a local file."""
    from datasets import load_dataset
    return load_dataset(repo_id, task)[split].to_pandas()


def _frame(data):
    """DataFrame | local table path | (repo_id, task[, split]) tuple as in the reference's CLI."""
    import pandas as pd
    if isinstance(data, pd.DataFrame):
        return data.reset_index(drop=True)
    if isinstance(data, (tuple, list)) and 2 <= len(data) <= 3 and all(isinstance(x, str) for x in data):
        return load_task(*data).reset_index(drop=True)
    path = str(data)
    if path.endswith(".parquet"):
        return pd.read_parquet(path)
    return pd.read_csv(path, sep="\t" if path.endswith((".tsv", ".txt")) else ",")


def _probs_or_infer(df_col, model, tokenizer, device, idx, batch_size, logits_path, save_logits):
    import pandas as pd
    if logits_path is not None:
        return pd.read_csv(logits_path, sep="\t").values
    if model is None:
        raise ValueError("either a model or logits_path is needed")
    probs = masked_probs(model, tokenizer, list(df_col), idx, device, batch_size)
    if save_logits and sharding.world()[0] == 0:
        pd.DataFrame(probs, columns=list(NUCLEOTIDES)).to_csv(save_logits, sep="\t", index=False)
    return probs


def _emit(metrics: Dict[str, float], metrics_json: Optional[str], extra: Optional[dict] = None) -> Dict[str, float]:
    if sharding.world()[0] != 0:            # under torchrun every rank holds the full probabilities and computes the (cheap) metrics;
        return metrics                      # rank 0 alone prints and writes
    for key, v in metrics.items():
        print(f"{key}\t{v:.6f}")
    if metrics_json:
        with open(metrics_json, "w") as f:
            json.dump({**{k.lower(): v for k, v in metrics.items()}, **(extra or {})}, f, indent=2)
    return metrics


def evo_cons(data, model=None, tokenizer=None, device="cuda:0", token_idx: int = 255, batch_size: int = 128,
             seq_column: str = "sequence", save_logits=None, logits_path=None, metrics_json=None) -> Dict[str, float]:
    """:324-372 - masked probabilities at one index; AUROC / AUPRC of the reference-base probability against `label`."""
    df = _frame(data)
    probs = _probs_or_infer(df[seq_column], model, tokenizer, device, token_idx, batch_size, logits_path, save_logits)
    assert probs.shape[0] == len(df), f"Row mismatch: probs={probs.shape[0]} examples={len(df)}"
    sc = refprob_scores(df[seq_column], probs, token_idx)
    y = df["label"].astype(int).to_numpy()
    return _emit({"AUROC": auroc(y, sc), "AUPRC": average_precision(y, sc)}, metrics_json, {"token_idx": token_idx})


def _positions(mask_idx, motif_len: int) -> List[int]:
    """The reference pairs the probabilities — which `_masked_probs` returns in ASCENDING position order (a masked_select
    over the sequence, :129-140) — with the true bases taken in the order the user gave `mask_idx`
    (`_compute_true_tokens_from_seq(df[seq], positions)`, :414 / :518).  The same pairing is kept here so results are
    identical on every input; with an unsorted `mask_idx` that pairing is inconsistent (in the reference too): warn."""
    positions = [int(x) for x in mask_idx]
    assert len(positions) == motif_len, "mask_idx count must equal motif_len"
    if positions != sorted(positions):
        logging.warning("mask_idx %s is not ascending: probabilities come out in ascending position order but true bases are "
                        "read in the given order (reference behaviour) - sort mask_idx", positions)
    return positions


def motif_acc(data, model=None, tokenizer=None, device="cuda:0", mask_idx: Sequence[int] = (255, 256, 257), motif_len: int = 3,
              batch_size: int = 128, seq_column: str = "sequence", save_logits=None, logits_path=None,
              metrics_json=None) -> Dict[str, float]:
    """:374-428 - multi-position masking; token and whole-motif accuracy."""
    df = _frame(data)
    positions = _positions(mask_idx, motif_len)
    probs = _probs_or_infer(df[seq_column], model, tokenizer, device, positions, batch_size, logits_path, save_logits)
    assert probs.shape[0] == len(df) * len(positions), f"Row mismatch: probs={probs.shape[0]} expected={len(df) * len(positions)}"
    tt = true_tokens(df[seq_column], positions)
    return _emit({"token_accuracy": token_accuracy(probs, tt), "motif_accuracy": motif_accuracy(probs, tt, motif_len)},
                 metrics_json)


def core_noncore(data, model=None, tokenizer=None, device="cuda:0", mask_idx: Sequence[int] = (255, 256, 257),
                 motif_len: int = 3, batch_size: int = 128, seq_column: str = "sequence", label_column: str = "label",
                 save_logits=None, logits_path=None, metrics_json=None) -> Dict[str, float]:
    """:478-530 - AUROC and AUPRC of the mean true-base probability over the masked positions against `label_column`
    (printed as `AUROC` / `AUPRC`, metrics_json keys `auroc` / `auprc`: :523-530)."""
    df = _frame(data)
    positions = _positions(mask_idx, motif_len)
    probs = _probs_or_infer(df[seq_column], model, tokenizer, device, positions, batch_size, logits_path, save_logits)
    assert probs.shape[0] == len(df) * len(positions)
    sc = avg_trueprob_scores(probs, true_tokens(df[seq_column], positions), motif_len)
    y = df[label_column].astype(int).to_numpy()
    return _emit({"AUROC": auroc(y, sc), "AUPRC": average_precision(y, sc)}, metrics_json)


def sv_effect(data, model, tokenizer, device="cuda:0", batch_size: int = 64, flanking: int = 5, output=None,
              save_ref_logits=None, save_mut_logits=None) -> Dict[str, float]:
    """:430-476 - un-masked probabilities of RefSeq and MutSeq, boundary LLR per row, AUPRC against `label`."""
    df = _frame(data)
    missing = [c for c in ("RefSeq", "MutSeq", "left", "right", "label") if c not in df.columns]
    if missing:
        raise KeyError(f"Missing required columns: {missing}")
    ref_p = unmasked_probs(df["RefSeq"], tokenizer, model, device, batch_size)
    mut_p = unmasked_probs(df["MutSeq"], tokenizer, model, device, batch_size)
    rank0 = sharding.world()[0] == 0
    if save_ref_logits and rank0:
        np.savez_compressed(save_ref_logits, logits=ref_p)
    if save_mut_logits and rank0:
        np.savez_compressed(save_mut_logits, logits=mut_p)
    scores = sv_llr_boundary(df["left"], df["right"], df["MutSeq"], ref_p, mut_p, flanking)
    res = _emit({"AUPRC": average_precision(df["label"].astype(int).to_numpy(), scores)}, None)
    if output and rank0:
        out = df.copy()
        out["score"] = scores
        out.drop(columns=["Left5_Positions", "Right5_Positions"], errors="ignore").to_csv(output, sep="\t", index=False)
    return res


# ---------------------------------------------------------------------------------------------------------------------
# command line: the reference's `python src/zero-shot-eval.py <sub-command> --repo_id ... --task ... --model ...`
# (Fire CLI, :533-534; flags as in docs/zero-shot-eval.md).  `--data <local table>` replaces --repo_id/--task when the
# table is a file; everything else keeps the reference's names and defaults.
def _load_model(model: str, device: str):
    """`_load_model` (:42-72): bf16 on the GPU; tokenizer from the same snapshot."""
    from .zero_shot import load_model_and_tokenizer
    return load_model_and_tokenizer(model, device)


def main(argv: Optional[Sequence[str]] = None):
    import argparse
    p = argparse.ArgumentParser(prog="plantcad2_eval", description="PlantCAD2 zero-shot evaluation tasks on the MI355X engine")
    sub = p.add_subparsers(dest="cmd", required=True)

    def common(q, batch_size):
        for f in ("repo_id", "task"):
            q.add_argument("--" + f, "--" + f.replace("_", "-"), dest=f, default=None)
        q.add_argument("--data", default=None, help="local .tsv/.csv/.parquet table instead of --repo_id/--task")
        q.add_argument("--split", default="valid")
        q.add_argument("--model", default="kuleshov-group/PlantCAD2-Small-l24-d0768")
        q.add_argument("--device", default="cuda:0")
        q.add_argument("--batch_size", "--batch-size", dest="batch_size", type=int, default=batch_size)

    def masked(q):
        q.add_argument("--seq_column", "--seq-column", dest="seq_column", default="sequence")
        for f in ("save_logits", "logits_path", "metrics_json"):
            q.add_argument("--" + f, "--" + f.replace("_", "-"), dest=f, default=None)

    def idx_list(v):
        return [int(x) for x in str(v).strip("[]() ").replace(" ", "").split(",") if x != ""]

    q = sub.add_parser("evo_cons"); common(q, 128); masked(q)
    q.add_argument("--token_idx", "--token-idx", dest="token_idx", type=int, default=255)
    for name in ("motif_acc", "core_noncore"):
        q = sub.add_parser(name); common(q, 128); masked(q)
        q.add_argument("--mask_idx", "--mask-idx", dest="mask_idx", type=idx_list, default=[255, 256, 257])
        q.add_argument("--motif_len", "--motif-len", dest="motif_len", type=int, default=3)
        if name == "core_noncore":
            q.add_argument("--label_column", "--label-column", dest="label_column", default="label")
    q = sub.add_parser("sv_effect"); common(q, 64)
    q.add_argument("--flanking", type=int, default=5)
    for f in ("output", "save_ref_logits", "save_mut_logits"):
        q.add_argument("--" + f, "--" + f.replace("_", "-"), dest=f, default=None)
    a = p.parse_args(argv)

    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    if a.data is None and not (a.repo_id and a.task):
        p.error("give --data <table> or --repo_id and --task")
    # under `torchrun --nproc-per-node N` the windows of every task are sharded over the N GPUs (masked_probs / unmasked_probs);
    # rank 0 prints the metric lines and writes the files
    a.device = sharding.init_from_env(a.device)
    try:
        data = a.data if a.data is not None else (a.repo_id, a.task, a.split)
        need_model = a.cmd == "sv_effect" or getattr(a, "logits_path", None) is None
        model, tok = _load_model(a.model, a.device) if need_model else (None, None)
        if a.cmd == "evo_cons":
            return evo_cons(data, model, tok, a.device, token_idx=a.token_idx, batch_size=a.batch_size, seq_column=a.seq_column,
                            save_logits=a.save_logits, logits_path=a.logits_path, metrics_json=a.metrics_json)
        if a.cmd == "motif_acc":
            return motif_acc(data, model, tok, a.device, mask_idx=a.mask_idx, motif_len=a.motif_len, batch_size=a.batch_size,
                             seq_column=a.seq_column, save_logits=a.save_logits, logits_path=a.logits_path, metrics_json=a.metrics_json)
        if a.cmd == "core_noncore":
            return core_noncore(data, model, tok, a.device, mask_idx=a.mask_idx, motif_len=a.motif_len, batch_size=a.batch_size,
                                seq_column=a.seq_column, label_column=a.label_column, save_logits=a.save_logits,
                                logits_path=a.logits_path, metrics_json=a.metrics_json)
        return sv_effect(data, model, tok, a.device, batch_size=a.batch_size, flanking=a.flanking, output=a.output,
                         save_ref_logits=a.save_ref_logits, save_mut_logits=a.save_mut_logits)
    finally:
        sharding.shutdown()


if __name__ == "__main__":
    main()
