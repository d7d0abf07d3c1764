"""`predict_XGBoost.py` end-to-end — host glue after the embedding path (reference `src/predict_XGBoost.py:19-67`,
`src/train_XGBoost.py:122-124`): averaged embeddings [N, d_model] -> pre-trained XGBoost classifier -> P(label = 1).

The embedding extraction is the accelerated path (`embeddings.extract_embeddings`); the classifier is a CPU tree
ensemble.  `xgboost` is not a dependency here: `XGBJsonClassifier` reads the XGBoost JSON model format that
`XGBClassifier.save_model("*.json")` writes (learner.gradient_booster.model.trees: left_children, right_children,
split_indices, split_conditions, default_left; leaf value in split_conditions; binary:logistic with base_score in
probability space) and evaluates it natively on all host cores (`hostlib.xgb_margin`, host/xgb_eval.c: C + OpenMP, the stand-in for
XGBoost's C++ predictor; `margin_numpy` is the same walk in vectorised numpy, kept as the cross-check and as the path when no C
compiler is available).  The 16 classifier JSONs of the reference are large blobs
that are absent from the reference checkout, so this reader is pinned by hand-built models in tests/test_xgb.py, not
by the real files.  Same flags, cache files (`<prefix>_embeddings.npz` / `<prefix>_chunk_<i>_embeddings.npz`, key
`test`) and output (`<prefix>_predictions.tsv`, columns label, prediction) as the reference script.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
from typing import Optional, Sequence

import numpy as np


class XGBJsonClassifier:
    def __init__(self):
        self.trees = []
        self.base_margin = 0.0
        self.n_features = 0

    def load_model(self, path: str):
        with open(path) as f:
            m = json.load(f)
        learner = m["learner"]
        obj = learner.get("objective", {}).get("name", "binary:logistic")
        if obj not in ("binary:logistic", "binary:logitraw"):
            raise ValueError(f"unsupported objective {obj}: the reference trains XGBClassifier on 0/1 labels")
        lp = learner["learner_model_param"]
        bs = str(lp.get("base_score", "0.5")).strip("[]")
        base_score = float(bs)
        self.n_features = int(lp.get("num_feature", 0))
        self.base_margin = float(np.log(base_score / (1.0 - base_score))) if obj == "binary:logistic" else base_score
        booster = learner["gradient_booster"]
        if booster.get("name", "gbtree") != "gbtree":
            raise ValueError("only gbtree boosters are supported")
        self.trees = []
        for ti, t in enumerate(booster["model"]["trees"]):
            tree = dict(
                left=np.asarray(t["left_children"], dtype=np.int64), right=np.asarray(t["right_children"], dtype=np.int64),
                feat=np.asarray(t["split_indices"], dtype=np.int64), cond=np.asarray(t["split_conditions"], dtype=np.float32),
                dleft=np.asarray(t["default_left"], dtype=bool))
            self._validate_tree(ti, tree)
            self.trees.append(tree)
        # the same arrays concatenated over trees: what the native evaluator walks (host/xgb_eval.c)
        sizes = [len(t["left"]) for t in self.trees]
        self._off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        cat = (lambda k, dt: np.concatenate([t[k] for t in self.trees]).astype(dt) if self.trees else np.zeros(0, dtype=dt))
        self._flat = (cat("left", np.int32), cat("right", np.int32), cat("feat", np.int32), cat("cond", np.float32), cat("dleft", np.uint8))
        return self

    def _validate_tree(self, ti: int, t):
        """Every node reachable from the root exactly once, children and split features in range: the native walk trusts this."""
        n = len(t["left"])
        if n == 0 or not (len(t["right"]) == len(t["feat"]) == len(t["cond"]) == len(t["dleft"]) == n):
            raise ValueError(f"tree {ti}: empty or ragged node arrays")
        seen = np.zeros(n, dtype=bool)
        stack = [0]
        while stack:
            k = stack.pop()
            if not 0 <= k < n or seen[k]:
                raise ValueError(f"tree {ti}: node {k} out of range or reached twice")
            seen[k] = True
            l, r = int(t["left"][k]), int(t["right"][k])
            if l == -1:
                continue
            f = int(t["feat"][k])
            if r == -1 or f < 0 or (self.n_features and f >= self.n_features):
                raise ValueError(f"tree {ti}: node {k} has one child or a split feature ({f}) outside [0, {self.n_features or 'n_features'})")
            stack += [l, r]

    def margin(self, X: np.ndarray) -> np.ndarray:
        """base margin + sum of leaf values, float64 [rows]: natively on all host cores when libpcad_host.so can be built / loaded
        (bit-identical to `margin_numpy`: same comparisons, same order of additions)."""
        X = np.asarray(X, dtype=np.float32)
        nf = self.n_features or (int(self._flat[2].max()) + 1 if len(self._flat[2]) else 0)
        if X.ndim != 2 or X.shape[1] < nf:
            raise ValueError(f"embeddings must be [rows, >= {nf}] (got {X.shape})")
        try:
            from . import hostlib
            hostlib.load_library()
        except Exception as ex:                       # no gcc / read-only tree: the numpy walk computes the same numbers
            logging.warning("libpcad_host.so unavailable (%s): evaluating the XGBoost ensemble in numpy", ex)
            return self.margin_numpy(X)
        return hostlib.xgb_margin(X, nf, self._off, *self._flat, self.base_margin)

    def margin_numpy(self, X: np.ndarray) -> np.ndarray:
        X = np.asarray(X, dtype=np.float32)
        out = np.full(X.shape[0], self.base_margin, dtype=np.float64)
        rows = np.arange(X.shape[0])
        for t in self.trees:
            node = np.zeros(X.shape[0], dtype=np.int64)
            active = t["left"][node] != -1
            while active.any():
                n = node[active]
                v = X[rows[active], t["feat"][n]]
                go_left = np.where(np.isnan(v), t["dleft"][n], v < t["cond"][n])      # xgboost: x < threshold -> left
                node[active] = np.where(go_left, t["left"][n], t["right"][n])
                active = t["left"][node] != -1
            out += t["cond"][node]                                                     # leaf value
        return out

    def predict_proba(self, X: np.ndarray) -> np.ndarray:
        p1 = 1.0 / (1.0 + np.exp(-self.margin(X)))
        return np.stack([1.0 - p1, p1], axis=1).astype(np.float32)


def infer_xgboost_model(model, embeddings) -> np.ndarray:
    logging.info("Inferencing XGBoost model")
    return model.predict_proba(embeddings)[:, 1]


def parse_args(argv: Optional[Sequence[str]] = None):
    p = argparse.ArgumentParser()
    p.add_argument("-test", type=str, help="The directory of test data")
    p.add_argument("-model", type=str, help="The directory of pre-trained model")
    p.add_argument("-classifier", type=str, help="The directory of trained XGBoost models")
    p.add_argument("-output", type=str, help="The directory of output")
    p.add_argument("-device", type=str, default="cuda:0", help="The device to run the model")
    p.add_argument("-batchSize", type=int, default=None,
                   help="The batch size for the model (default 128 as in the reference, raised to the engine's preferred batch; a "
                        "value given here is used as is)")
    p.add_argument("-tokenIdx", type=int, default=255, help="The index of the nucleotide")
    p.add_argument("-save_memory", action="store_true", help="Flag to save memory, it only works for testing")
    p.add_argument("-chunk_size", type=int, default=100000, help="The chunk size for testing, with -save_memory")
    args = p.parse_args(argv)
    args.batchExplicit = args.batchSize is not None
    if args.batchSize is None:
        args.batchSize = 128
    return args


def main(argv: Optional[Sequence[str]] = None):
    import pandas as pd
    from . import sharding
    from .embeddings import extract_embeddings, load_data
    from .zero_shot import load_model_and_tokenizer
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    args = parse_args(argv)
    args.device = sharding.init_from_env(args.device)
    os.makedirs(args.output, exist_ok=True)
    model, tokenizer = load_model_and_tokenizer(args.model, args.device)
    test_sequences, test_labels = load_data(args.test)
    clf = XGBJsonClassifier().load_model(args.classifier)
    prefix = os.path.basename(args.test).split(".")[0]
    rank, _ = sharding.world()

    from concurrent.futures import ThreadPoolExecutor
    from .embeddings import save_embedding_cache
    writer = ThreadPoolExecutor(max_workers=1)          # cache files are written (deflate, one core) while the next chunk runs
    pending = []

    def embeddings_for(seqs, cache):
        # rank 0 looks, every rank follows (the other branch is a collective); only rank 0 needs the values
        if sharding.rank0_decides(os.path.exists(cache), args.device):
            if rank != 0:
                return None
            logging.info(f"Found pre-computed embeddings, loading from file {cache}")
            return np.load(cache)["test"]
        emb = extract_embeddings(model, seqs, args.device, args.tokenIdx, tokenizer, args.batchSize, args.batchExplicit)
        if rank == 0:
            pending.append(writer.submit(save_embedding_cache, cache, test=emb))
        return emb

    try:
        if args.save_memory:
            preds = []
            for i in range(0, len(test_sequences), args.chunk_size):
                emb = embeddings_for(test_sequences[i:i + args.chunk_size], os.path.join(args.output, f"{prefix}_chunk_{i}_embeddings.npz"))
                if rank == 0:                                # the classifier step is host work on rank 0 (the only writer below)
                    preds.append(infer_xgboost_model(clf, emb))
            predictions = np.concatenate(preds, axis=0) if preds else np.zeros(0, dtype=np.float32)
        else:
            emb = embeddings_for(test_sequences, os.path.join(args.output, prefix + "_embeddings.npz"))
            predictions = infer_xgboost_model(clf, emb) if rank == 0 else None
        # every rank leaves the process group after the last all-gather (rank 0's table writing below is host work; see
        # zero_shot.main)
        sharding.shutdown()
        for fut in pending:
            fut.result()                                     # surface write errors; the files are complete before the run ends
    finally:
        writer.shutdown(wait=True)
    if rank == 0:
        pd.DataFrame({"label": test_labels, "prediction": predictions}).to_csv(
            os.path.join(args.output, f"{prefix}_predictions.tsv"), sep="\t", index=False)
        logging.info(f"Saved predictions to {os.path.join(args.output, f'{prefix}_predictions.tsv')}")


if __name__ == "__main__":
    main()
