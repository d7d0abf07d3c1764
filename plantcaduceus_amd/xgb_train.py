"""`train_XGBoost.py` end-to-end (reference `src/train_XGBoost.py:13-27,116-270`): averaged embeddings of the train / validation /
test tables from the accelerated path, the caches and result files under the reference's names, the classifier step.

What runs where:
  * embeddings       `embeddings.extract_embeddings` (the HIP engine; sharded over the ranks of a torchrun launch, one all-gather per
                     table / chunk) -> `train_valid_embeddings.npz` (keys `train`, `valid`), `<prefix>_embeddings.npz` /
                     `<prefix>_chunk_<i>_embeddings.npz` (key `test`): found caches are reused exactly as the reference does;
  * gradient boosting itself (`XGBClassifier(n_estimators=1000, max_depth=6, learning_rate=0.1, random_state=seed).fit`,
                     src/train_XGBoost.py:116-120) is the third-party trainer: imported when the environment has `xgboost`, and asked for
                     only AFTER the embeddings are cached, so a box without it still produces everything the GPU is needed for
                     (the saved `seed_<seed>_XGBoost.json` of any XGBoost run is then picked up as the reference picks it up);
  * inference        an existing `seed_<seed>_XGBoost.json` is evaluated natively (`xgb_predict.XGBJsonClassifier`, host/xgb_eval.c):
                     `-test_only` and re-runs need no xgboost at all;
  * metrics          ROC AUC / PR AUC (`sklearn.metrics.auc(roc_curve)` / `average_precision_score` semantics: mid-rank AUROC,
                     step-wise average precision - `plantcad2_eval.auroc` / `average_precision`), written as the reference's
                     `seed_<seed>_<prefix>_metrics.txt` (and the two-panel `.png` when matplotlib is importable),
                     predictions as `seed_<seed>_<prefix>_predictions.npz` (key `predictions`).
Rank 0 writes; every rank leaves the process group after the last all-gather.
"""
from __future__ import annotations

import argparse
import logging
import os
from typing import Optional, Sequence

import numpy as np

from .xgb_predict import XGBJsonClassifier, infer_xgboost_model

XGB_PARAMS = dict(n_estimators=1000, max_depth=6, learning_rate=0.1, n_jobs=-1)       # src/train_XGBoost.py:118


def parse_args(argv: Optional[Sequence[str]] = None):
    p = argparse.ArgumentParser()
    p.add_argument("-train", type=str, help="The directory of training data")
    p.add_argument("-valid", type=str, help="The directory of validation data")
    p.add_argument("-test", type=str, help="The directory of test data")
    p.add_argument("-model", type=str, help="The directory of pre-trained model")
    p.add_argument("-output", type=str, help="The directory of output")
    p.add_argument("-device", type=str, default="cuda:0", help="The device to run the model")
    p.add_argument("-batchSize", type=int, default=None,
                   help="The batch size for the model (default 128 as in the reference, raised to the engine's preferred batch; a "
                        "value given here is used as is)")
    p.add_argument("-tokenIdx", type=int, default=255, help="The index of the nucleotide")
    p.add_argument("-test_only", action="store_true", help="Flag to perform only testing")
    p.add_argument("-save_memory", action="store_true", help="Flag to save memory, it only works for testing")
    p.add_argument("-chunk_size", type=int, default=100000, help="The chunk size for testing, with -save_memory")
    p.add_argument("-seed", type=int, default=42, help="The random seed to train XGBoost model")
    args = p.parse_args(argv)
    args.batchExplicit = args.batchSize is not None
    if args.batchSize is None:
        args.batchSize = 128
    return args


def train_xgboost_model(train_embeddings, train_labels, valid_embeddings, valid_labels, random_state: int = 42):
    """The reference's trainer call (src/train_XGBoost.py:116-120).  `xgboost` is the reference's own dependency, not this repo's."""
    try:
        import xgboost as xgb
    except ImportError as ex:
        raise RuntimeError(
            "training the classifier needs the `xgboost` package (the reference's trainer, src/train_XGBoost.py:118), which is not "
            "installed here.  The embeddings are cached in the output directory: run this command again where xgboost is available "
            "(no GPU needed for that step), or place a trained seed_<seed>_XGBoost.json there.") from ex
    logging.info("Training XGBoost model")
    model = xgb.XGBClassifier(random_state=random_state, **XGB_PARAMS)
    model.fit(train_embeddings, train_labels, eval_set=[(valid_embeddings, valid_labels)])
    return model


def evaluate_model(predictions, labels):
    """-> (roc_auc, prauc) with sklearn's definitions (src/train_XGBoost.py:126-132)."""
    from .plantcad2_eval import auroc, average_precision
    return float(auroc(labels, predictions)), float(average_precision(labels, predictions))


def _curves(predictions, labels):
    """ROC and precision-recall points for the plot (thresholds at every distinct score, descending)."""
    y = np.asarray(labels, dtype=np.float64)
    s = np.asarray(predictions, dtype=np.float64)
    order = np.argsort(-s, kind="stable")
    y, s = y[order], s[order]
    last = np.r_[np.nonzero(np.diff(s))[0], len(s) - 1]           # last index of every run of equal scores
    tp = np.cumsum(y)[last]
    fp = (last + 1) - tp
    P, N = max(y.sum(), 1.0), max(len(y) - y.sum(), 1.0)
    fpr, tpr = np.r_[0.0, fp / N], np.r_[0.0, tp / P]
    precision, recall = np.r_[(tp / (tp + fp))[::-1], 1.0], np.r_[(tp / P)[::-1], 0.0]     # recall descending, ending at (0, 1)
    return fpr, tpr, precision, recall


def write_metrics(predictions, labels, output_dir: str, prefix: str, random_state: int):
    """`seed_<seed>_<prefix>_metrics.txt` (two lines, two decimals: src/train_XGBoost.py:152-155) and, when matplotlib is there, the
    two-panel `seed_<seed>_<prefix>_metrics.png`."""
    roc_auc, prauc = evaluate_model(predictions, labels)
    base = os.path.join(output_dir, f"seed_{random_state}_{prefix}_metrics")
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fpr, tpr, precision, recall = _curves(predictions, labels)
        fig, axs = plt.subplots(1, 2, figsize=(12, 6))
        axs[0].plot(fpr, tpr, label=f"AUC = {roc_auc:.2f}", linewidth=2)
        axs[0].set_title("ROC Curve"); axs[0].set_xlabel("False Positive Rate"); axs[0].set_ylabel("True Positive Rate")
        axs[0].legend(loc="lower right")
        axs[1].plot(recall, precision, label=f"PRAUC = {prauc:.2f}", linewidth=2)
        axs[1].set_title("Precision-Recall Curve"); axs[1].set_xlabel("Recall"); axs[1].set_ylabel("Precision")
        axs[1].legend(loc="lower left")
        plt.tight_layout()
        plt.savefig(base + ".png")
        plt.close(fig)
    except ImportError:
        logging.info("matplotlib not available: metrics written as text only")
    with open(base + ".txt", "w") as f:
        f.write(f"ROC AUC: {roc_auc:.2f}\n")
        f.write(f"PRAUC: {prauc:.2f}\n")
    return roc_auc, prauc


def main(argv: Optional[Sequence[str]] = None):
    from . import sharding
    from .embeddings import extract_embeddings, load_data, save_embedding_cache
    from .zero_shot import load_model_and_tokenizer
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    args = parse_args(argv)
    if args.test_only and not args.test:
        logging.error("Please provide the test data")
        return
    args.device = sharding.init_from_env(args.device)
    os.makedirs(args.output, exist_ok=True)
    out = lambda name: os.path.join(args.output, name)                                   # noqa: E731
    model_json = out(f"seed_{args.seed}_XGBoost.json")
    rank, _ = sharding.world()
    if args.test_only:
        # the reference fails at xgb_model.load_model before any forward (src/train_XGBoost.py:173): so does every rank here
        if sharding.rank0_decides(not os.path.exists(model_json), args.device):
            sharding.shutdown()
            raise FileNotFoundError(f"{model_json} not found: -test_only evaluates a classifier trained by an earlier run")
        # a malformed JSON also fails now, not after the embeddings - on EVERY rank: only rank 0 parses the file, its verdict is
        # broadcast (a rank 0 that raised alone would leave the others waiting in load_model / the all-gathers until torchrun's
        # watchdog tears them down)
        err = None
        if rank == 0:
            try:
                XGBJsonClassifier().load_model(model_json)
            except Exception as ex:                             # any reader failure: bad JSON, missing keys, unsupported booster
                err = ex
        if sharding.rank0_decides(err is not None, args.device):
            sharding.shutdown()
            raise ValueError(f"{model_json} is not a classifier this evaluator reads" + (f": {err!r}" if err is not None else " (rank 0 reported the error)"))
    model, tokenizer = load_model_and_tokenizer(args.model, args.device)

    def embed(seqs):
        return extract_embeddings(model, seqs, args.device, args.tokenIdx, tokenizer, args.batchSize, args.batchExplicit)

    # ---- every forward first (collectives: all ranks), host-only work afterwards (rank 0) --------------------------------
    train_labels = valid_labels = train_emb = valid_emb = None
    if not args.test_only:
        train_sequences, train_labels = load_data(args.train)
        valid_sequences, valid_labels = load_data(args.valid)
        cache = out("train_valid_embeddings.npz")
        # rank 0 looks, every rank follows: the else-branch is a collective (all-gather), so all ranks must take the same one even
        # when they do not see the same output directory (node-local -output, stale attribute caches); only rank 0 reads the file
        if sharding.rank0_decides(os.path.exists(cache), args.device):
            if rank == 0:
                logging.info(f"Found pre-computed embeddings, loading from file {cache}")
                z = np.load(cache)
                train_emb, valid_emb = z["train"], z["valid"]
        else:
            train_emb, valid_emb = embed(train_sequences), embed(valid_sequences)
            if rank == 0:
                save_embedding_cache(cache, train=train_emb, valid=valid_emb)
    test_labels, test_chunks, prefix = None, [], None
    if args.test:
        test_sequences, test_labels = load_data(args.test)
        prefix = os.path.basename(args.test).split(".")[0]
        spans = ([(i, f"{prefix}_chunk_{i}_embeddings.npz") for i in range(0, len(test_sequences), args.chunk_size)]
                 if args.save_memory else [(0, prefix + "_embeddings.npz")])
        if args.save_memory:
            logging.info(f"Saving memory by splitting the test data into smaller chunks with size {args.chunk_size}")
        for i, name in spans:
            cache = out(name)
            # -save_memory: a chunk's embeddings live in its cache file only (re-read one at a time at the classifier step below)
            if sharding.rank0_decides(os.path.exists(cache), args.device):
                if rank == 0:
                    logging.info(f"Found pre-computed embeddings, loading from file {cache}")
                    test_chunks.append(cache if args.save_memory else np.load(cache)["test"])
                continue
            emb = embed(test_sequences[i:i + args.chunk_size] if args.save_memory else test_sequences)
            if rank == 0:
                save_embedding_cache(cache, test=emb)
            test_chunks.append(cache if args.save_memory else emb)
    sharding.shutdown()
    if rank != 0:
        return

    # ---- classifier: found JSON (native evaluator) or the reference's trainer ---------------------------------------------
    if os.path.exists(model_json):
        if not args.test_only:
            logging.info(f"Found pre-trained XGBoost model, loading from file {model_json}")
        clf = XGBJsonClassifier().load_model(model_json)
    elif args.test_only:
        raise FileNotFoundError(f"{model_json} not found: -test_only evaluates a classifier trained by an earlier run")
    else:
        xgb_model = train_xgboost_model(train_emb, train_labels, valid_emb, valid_labels, random_state=args.seed)
        xgb_model.save_model(model_json)
        valid_predictions = infer_xgboost_model(xgb_model, valid_emb)
        np.savez_compressed(out(f"seed_{args.seed}_valid_predictions.npz"), predictions=valid_predictions)
        write_metrics(valid_predictions, valid_labels, args.output, os.path.basename(args.valid).split(".")[0], args.seed)
        clf = XGBJsonClassifier().load_model(model_json)          # the test tables go through the saved file, as in the reference (:218-220)
    if args.test:
        preds = [infer_xgboost_model(clf, np.load(c)["test"] if isinstance(c, str) else c) for c in test_chunks]
        predictions = np.concatenate(preds, axis=0) if preds else np.zeros(0, dtype=np.float32)
        np.savez_compressed(out(f"seed_{args.seed}_{prefix}_predictions.npz"), predictions=predictions)
        roc_auc, prauc = write_metrics(predictions, test_labels, args.output, prefix, args.seed)
        logging.info(f"{prefix}: ROC AUC {roc_auc:.4f}, PRAUC {prauc:.4f}")


if __name__ == "__main__":
    main()
