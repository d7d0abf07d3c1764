"""CPU: the XGBoost-JSON reader / evaluator and the predict_XGBoost-compatible CLI plumbing (oracle stand-in model)."""
import json
import os

import numpy as np
import pandas as pd

from oracle import caduceus_oracle as O
from plantcaduceus_amd import xgb_predict, zero_shot
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer


def _tree(left, right, feat, cond, dleft):
    return dict(left_children=left, right_children=right, split_indices=feat, split_conditions=cond, default_left=dleft,
                base_weights=[0.0] * len(left), parents=[2147483647] + [0] * (len(left) - 1))


def _model_json(trees, n_features, base_score="5E-1"):
    return {"learner": {"objective": {"name": "binary:logistic"},
                        "learner_model_param": {"base_score": base_score, "num_class": "0", "num_feature": str(n_features)},
                        "gradient_booster": {"name": "gbtree", "model": {"trees": trees, "tree_info": [0] * len(trees)}}},
            "version": [2, 0, 3]}


def test_xgb_json_evaluator(tmp_path):
    # tree 0: x[0] < 0.5 ? (x[2] < -1 ? 0.3 : -0.2) : 0.7 ;  tree 1: x[1] < 2 ? -0.5 : 0.25 (missing -> right)
    t0 = _tree([1, 3, -1, -1, -1], [2, 4, -1, -1, -1], [0, 2, 0, 0, 0], [0.5, -1.0, 0.7, 0.3, -0.2], [1, 1, 0, 0, 0])
    t1 = _tree([1, -1, -1], [2, -1, -1], [1, 0, 0], [2.0, -0.5, 0.25], [0, 0, 0])
    path = tmp_path / "m.json"
    json.dump(_model_json([t0, t1], 3, base_score="2.5E-1"), open(path, "w"))
    clf = xgb_predict.XGBJsonClassifier().load_model(str(path))
    X = np.array([[0.0, 0.0, -2.0], [0.0, 5.0, 0.0], [1.0, np.nan, 0.0], [np.nan, 1.0, -5.0], [0.5, 2.0, 0.0]], dtype=np.float32)
    base = np.log(0.25 / 0.75)
    want = base + np.array([0.3 - 0.5, -0.2 + 0.25, 0.7 + 0.25, 0.3 - 0.5, 0.7 + 0.25])
    np.testing.assert_allclose(clf.margin(X), want, rtol=1e-6)
    p = clf.predict_proba(X)
    np.testing.assert_allclose(p[:, 1], 1 / (1 + np.exp(-want)), rtol=1e-6)
    np.testing.assert_allclose(p.sum(1), 1.0, rtol=1e-6)
    np.testing.assert_allclose(xgb_predict.infer_xgboost_model(clf, X), p[:, 1])


def test_predict_cli_plumbing(tmp_path, monkeypatch, golden_dir):
    cfg = make_config("x", d_model=64, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=1), cfg))
    model.config = cfg
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (model, CaduceusTokenizer()))
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:5]
    test = tmp_path / "mytest.tsv"
    pd.DataFrame({"sequences": src["sequences"], "label": [0, 1, 0, 1, 1]}).to_csv(test, sep="\t", index=False)
    t = _tree([1, -1, -1], [2, -1, -1], [3, 0, 0], [0.0, -1.0, 1.0], [1, 0, 0])
    clf = tmp_path / "clf.json"
    json.dump(_model_json([t], 64), open(clf, "w"))
    out = tmp_path / "out"
    argv = ["-test", str(test), "-model", "unused", "-classifier", str(clf), "-output", str(out), "-device", "cpu", "-batchSize", "2"]
    xgb_predict.main(argv)
    res = pd.read_csv(out / "mytest_predictions.tsv", delimiter="\t")
    assert list(res.columns) == ["label", "prediction"] and len(res) == 5
    emb = np.load(out / "mytest_embeddings.npz")["test"]
    assert emb.shape == (5, 64)
    want = 1 / (1 + np.exp(-np.where(emb[:, 3] < 0.0, -1.0, 1.0)))
    np.testing.assert_allclose(res["prediction"].to_numpy(), want, rtol=1e-5)
    # second run takes the cached embeddings (model must not be called), chunked mode writes per-chunk caches
    monkeypatch.setattr(model, "forward", lambda *a, **k: (_ for _ in ()).throw(AssertionError("cache not used")))
    xgb_predict.main(argv)
    monkeypatch.undo()
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (model, CaduceusTokenizer()))
    xgb_predict.main(["-test", str(test), "-model", "unused", "-classifier", str(clf), "-output", str(tmp_path / "out2"),
                      "-device", "cpu", "-save_memory", "-chunk_size", "2"])
    assert sorted(os.listdir(tmp_path / "out2")) == ["mytest_chunk_0_embeddings.npz", "mytest_chunk_2_embeddings.npz",
                                                     "mytest_chunk_4_embeddings.npz", "mytest_predictions.tsv"]
    res2 = pd.read_csv(tmp_path / "out2" / "mytest_predictions.tsv", delimiter="\t")
    np.testing.assert_allclose(res2["prediction"].to_numpy(), want, rtol=1e-5)


def _random_tree(rng, n_features, depth):
    """random binary tree in XGBoost's JSON arrays (node 0 = root, children appended breadth-first)"""
    left, right, feat, cond, dleft = [-1], [-1], [0], [float(rng.normal())], [0]
    frontier = [(0, 0)]
    while frontier:
        node, d = frontier.pop(0)
        if d < depth and rng.random() < 0.8:
            l, r = len(left), len(left) + 1
            for _ in range(2):
                left.append(-1); right.append(-1); feat.append(0); cond.append(float(rng.normal() * 0.3)); dleft.append(0)
            left[node], right[node] = l, r
            feat[node], cond[node], dleft[node] = int(rng.integers(0, n_features)), float(rng.normal()), int(rng.integers(0, 2))
            frontier += [(l, d + 1), (r, d + 1)]
    return _tree(left, right, feat, cond, dleft)


def test_xgb_evaluator_vs_scalar_walk_on_random_forests(tmp_path):
    """the vectorised evaluator against a plain per-row recursive walk of the same JSON (x < threshold -> left, missing ->
    default_left, leaf value in split_conditions, margins summed over trees + logit(base_score)) on 40 random trees with
    missing values."""
    rng = np.random.default_rng(5)
    F = 24
    trees = [_random_tree(rng, F, depth=int(rng.integers(1, 7))) for _ in range(40)]
    path = tmp_path / "forest.json"
    json.dump(_model_json(trees, F, base_score="3.1E-1"), open(path, "w"))
    clf = xgb_predict.XGBJsonClassifier().load_model(str(path))
    X = rng.normal(size=(300, F)).astype(np.float32)
    X[rng.random(X.shape) < 0.1] = np.nan

    def walk(t, x):
        n = 0
        while t["left_children"][n] != -1:
            v = x[t["split_indices"][n]]
            go_left = bool(t["default_left"][n]) if np.isnan(v) else bool(v < np.float32(t["split_conditions"][n]))
            n = t["left_children"][n] if go_left else t["right_children"][n]
        return np.float32(t["split_conditions"][n])
    want = np.array([np.log(0.31 / 0.69) + sum(float(walk(t, x)) for t in trees) for x in X])
    np.testing.assert_allclose(clf.margin(X), want, rtol=1e-5, atol=1e-6)
    assert len({round(v, 3) for v in want}) > 100                      # the forest really separates rows


import pytest  # noqa: E402


def test_native_evaluator_equals_numpy_walk_bit_for_bit(tmp_path):
    """libpcad_host.so (host/xgb_eval.c, C + OpenMP: what `margin` runs) against `margin_numpy` on random forests with missing
    values, wider-than-needed feature rows, a single-leaf tree, zero rows; bit-identical float64 margins."""
    from plantcaduceus_amd import hostlib
    assert hostlib.load_library().pcad_host_version() >= 1
    rng = np.random.default_rng(11)
    F = 40
    trees = [_random_tree(rng, F, depth=int(rng.integers(0, 7))) for _ in range(120)]
    trees.append(_tree([-1], [-1], [0], [0.25], [0]))                      # a stump that is a single leaf
    path = tmp_path / "forest.json"
    json.dump(_model_json(trees, F, base_score="7.3E-1"), open(path, "w"))
    clf = xgb_predict.XGBJsonClassifier().load_model(str(path))
    X = rng.normal(size=(1000, F + 3)).astype(np.float32)                  # callers may pass wider rows
    X[rng.random(X.shape) < 0.15] = np.nan
    a, b = clf.margin(X), clf.margin_numpy(X)
    assert a.dtype == np.float64 and np.array_equal(a, b)
    assert clf.margin(X[:0]).shape == (0,)
    np.testing.assert_array_equal(clf.margin(X[:65]), b[:65])              # one full row block + 1
    with pytest.raises(ValueError):
        clf.margin(X[:, :F - 1])                                           # fewer features than the model splits on


def test_malformed_trees_are_rejected_at_load(tmp_path):
    """the native walk trusts the arrays, so load_model validates them: child index out of range, a cycle, a split feature
    outside num_feature."""
    good = _tree([1, -1, -1], [2, -1, -1], [0, 0, 0], [0.5, 1.0, 2.0], [0, 0, 0])
    for bad in (dict(good, left_children=[5, -1, -1]), dict(good, right_children=[0, -1, -1]),
                dict(good, split_indices=[9, 0, 0]), dict(good, right_children=[-1, -1, -1])):
        path = tmp_path / "bad.json"
        json.dump(_model_json([bad], 4), open(path, "w"))
        with pytest.raises(ValueError):
            xgb_predict.XGBJsonClassifier().load_model(str(path))
    json.dump(_model_json([good], 4), open(tmp_path / "good.json", "w"))
    xgb_predict.XGBJsonClassifier().load_model(str(tmp_path / "good.json"))
    # a file WITHOUT num_feature (0): the width is derived from the trees, so a negative split index must still be refused at load,
    # and the native entry point refuses out-of-range features by itself (it must not trust its caller)
    neg = _model_json([dict(good, split_indices=[-3, 0, 0])], 0)
    json.dump(neg, open(tmp_path / "neg.json", "w"))
    with pytest.raises(ValueError):
        xgb_predict.XGBJsonClassifier().load_model(str(tmp_path / "neg.json"))
    from plantcaduceus_amd import hostlib
    X = np.zeros((3, 4), dtype=np.float32)
    off, L, R, C, Dl = np.array([0, 3]), np.array([1, -1, -1]), np.array([2, -1, -1]), np.array([0.5, 1.0, 2.0]), np.array([0, 0, 0])
    assert hostlib.xgb_margin(X, 4, off, L, R, np.array([3, 0, 0]), C, Dl, 0.0).shape == (3,)
    for bad_feat in ([-1, 0, 0], [4, 0, 0]):
        with pytest.raises(RuntimeError):
            hostlib.xgb_margin(X, 4, off, L, R, np.array(bad_feat), C, Dl, 0.0)




class _StubXGBClassifier:
    """stands in for xgboost.XGBClassifier in the train-CLI test: `fit` "learns" one stump on the feature whose class means differ
    most, `save_model` writes it in XGBoost's JSON format, `predict_proba` evaluates the same stump."""
    last_params = None

    def __init__(self, **params):
        type(self).last_params = params

    def fit(self, X, y, eval_set=None):
        X, y = np.asarray(X, dtype=np.float32), np.asarray(y)
        assert eval_set is not None and len(eval_set) == 1 and len(eval_set[0]) == 2
        d = X[y == 1].mean(0) - X[y == 0].mean(0)
        self.f = int(np.abs(d).argmax())
        self.thr = float(np.float32((X[y == 1, self.f].mean() + X[y == 0, self.f].mean()) / 2))
        self.lo, self.hi = (-2.0, 2.0) if d[self.f] > 0 else (2.0, -2.0)
        self.nf = X.shape[1]
        return self

    def save_model(self, path):
        t = _tree([1, -1, -1], [2, -1, -1], [self.f, 0, 0], [self.thr, self.lo, self.hi], [0, 0, 0])
        json.dump(_model_json([t], self.nf), open(path, "w"))

    def predict_proba(self, X):
        m = np.where(np.asarray(X, dtype=np.float32)[:, self.f] < np.float32(self.thr), self.lo, self.hi)
        p = 1 / (1 + np.exp(-m))
        return np.stack([1 - p, p], 1)


def test_train_cli_plumbing(tmp_path, monkeypatch, golden_dir):
    """reference src/train_XGBoost.py main() flow with its file names: train / valid embeddings cache, trainer call with the reference's
    hyper-parameters (xgboost stubbed: it is not installed here), saved JSON re-read by the native evaluator for the test table,
    predictions / metrics files, the -test_only re-run that needs neither trainer nor model forward, and the error path when
    xgboost is missing AFTER the caches were written."""
    import sys
    import types
    from plantcaduceus_amd import xgb_train
    cfg = make_config("x", d_model=64, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=1), cfg))
    model.config = cfg
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (model, CaduceusTokenizer()))
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")
    paths = {}
    for name, rows, labels in (("tr", slice(0, 12), [0, 1] * 6), ("va", slice(12, 18), [1, 0] * 3), ("te", slice(18, 23), [0, 1, 1, 0, 1])):
        paths[name] = tmp_path / f"{name}.tsv"
        pd.DataFrame({"sequences": src["sequences"].iloc[rows], "label": labels}).to_csv(paths[name], sep="\t", index=False)
    out = tmp_path / "out"
    argv = ["-train", str(paths["tr"]), "-valid", str(paths["va"]), "-test", str(paths["te"]), "-model", "unused", "-output", str(out),
            "-device", "cpu", "-batchSize", "4", "-seed", "7"]
    # (1) no xgboost: the embeddings are cached first, then a clear error
    monkeypatch.setitem(sys.modules, "xgboost", None)
    with pytest.raises(RuntimeError, match="xgboost"):
        xgb_train.main(argv)
    z = np.load(out / "train_valid_embeddings.npz")
    assert z["train"].shape == (12, 64) and z["valid"].shape == (6, 64)
    assert (out / "te_embeddings.npz").exists() and not (out / "seed_7_XGBoost.json").exists()
    # (2) with a trainer: caches reused (the model must not run again), everything written under the reference's names
    stub = types.ModuleType("xgboost")
    stub.XGBClassifier = _StubXGBClassifier
    monkeypatch.setitem(sys.modules, "xgboost", stub)
    monkeypatch.setattr(model, "forward", lambda *a, **k: (_ for _ in ()).throw(AssertionError("cache not used")))
    xgb_train.main(argv)
    assert _StubXGBClassifier.last_params == dict(n_estimators=1000, max_depth=6, learning_rate=0.1, n_jobs=-1, random_state=7)
    assert sorted(os.listdir(out)) == sorted(
        ["train_valid_embeddings.npz", "te_embeddings.npz", "seed_7_XGBoost.json", "seed_7_valid_predictions.npz",
         "seed_7_va_metrics.txt", "seed_7_va_metrics.png", "seed_7_te_predictions.npz", "seed_7_te_metrics.txt", "seed_7_te_metrics.png"])
    clf = xgb_predict.XGBJsonClassifier().load_model(str(out / "seed_7_XGBoost.json"))
    want = clf.predict_proba(np.load(out / "te_embeddings.npz")["test"])[:, 1]
    got = np.load(out / "seed_7_te_predictions.npz")["predictions"]
    np.testing.assert_allclose(got, want, rtol=1e-6)
    txt = open(out / "seed_7_te_metrics.txt").read().splitlines()
    assert txt[0].startswith("ROC AUC: ") and txt[1].startswith("PRAUC: ") and len(txt[0].split(": ")[1]) == 4
    # (3) -test_only, chunked: no trainer, no training tables; predictions equal the unchunked run's
    monkeypatch.setitem(sys.modules, "xgboost", None)
    monkeypatch.undo()
    monkeypatch.setattr(zero_shot, "load_model_and_tokenizer", lambda d, dev: (model, CaduceusTokenizer()))
    os.remove(out / "te_embeddings.npz")
    xgb_train.main(["-test", str(paths["te"]), "-model", "unused", "-output", str(out), "-device", "cpu", "-seed", "7", "-test_only",
                    "-save_memory", "-chunk_size", "2"])
    assert all((out / f"te_chunk_{i}_embeddings.npz").exists() for i in (0, 2, 4))
    np.testing.assert_allclose(np.load(out / "seed_7_te_predictions.npz")["predictions"], want, rtol=1e-6)
    with pytest.raises(FileNotFoundError):
        xgb_train.main(["-test", str(paths["te"]), "-model", "unused", "-output", str(tmp_path / "empty"), "-device", "cpu", "-test_only"])
    # a classifier file that does not parse fails before any forward, on every rank (rank 0 parses, the verdict is broadcast)
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "seed_42_XGBoost.json").write_text("{ not json")
    with pytest.raises(ValueError, match="not a classifier"):
        xgb_train.main(["-test", str(paths["te"]), "-model", "unused", "-output", str(bad), "-device", "cpu", "-test_only"])


def test_train_cli_metrics_follow_sklearn():
    """ROC AUC / PRAUC of evaluate_model against sklearn.metrics (what the reference calls), ties included."""
    sk = pytest.importorskip("sklearn.metrics")
    from plantcaduceus_amd import xgb_train
    rng = np.random.default_rng(3)
    y = rng.integers(0, 2, size=400)
    s = np.round(rng.random(400) * 0.6 + 0.3 * y, 2)                      # many tied scores
    roc_auc, prauc = xgb_train.evaluate_model(s, y)
    fpr, tpr, _ = sk.roc_curve(y, s)
    assert abs(roc_auc - sk.auc(fpr, tpr)) < 1e-12 and abs(prauc - sk.average_precision_score(y, s)) < 1e-12
    f2, t2, p2, r2 = xgb_train._curves(s, y)                             # the plotted points: every distinct threshold
    fpr_all, tpr_all, _ = sk.roc_curve(y, s, drop_intermediate=False)
    pr, rc, _ = sk.precision_recall_curve(y, s)
    assert np.allclose(f2, fpr_all) and np.allclose(t2, tpr_all)
    assert np.allclose(p2, pr) and np.allclose(r2, rc)


@pytest.mark.gpu
def test_predict_cli_on_gpu_from_snapshot(tmp_path, golden_dir):
    """reference src/predict_XGBoost.py:28-67 end to end on the GPU, nothing mocked: snapshot directory -> embeddings through
    the HIP path -> XGBoost-JSON classifier -> <prefix>_predictions.tsv; embeddings against the oracle on the same checkpoint,
    predictions against the classifier applied to them."""
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    d = str(tmp_path / "snap")
    cfg, sd = make_synthetic_checkpoint(d, "x", seed=21, stress=False, d_model=128, n_layer=2)
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:20]
    test = tmp_path / "gpu_test.tsv"
    rng = np.random.default_rng(0)
    pd.DataFrame({"sequences": src["sequences"], "label": rng.integers(0, 2, size=len(src))}).to_csv(test, sep="\t", index=False)
    trees = [_random_tree(rng, cfg.d_model, 4) for _ in range(12)]
    for t in trees:                                                      # thresholds on the scale of the embeddings
        t["split_conditions"] = [c * 0.05 if l != -1 else c for c, l in zip(t["split_conditions"], t["left_children"])]
    clf = tmp_path / "clf.json"
    json.dump(_model_json(trees, cfg.d_model), open(clf, "w"))
    out = tmp_path / "out"
    xgb_predict.main(["-test", str(test), "-model", d, "-classifier", str(clf), "-output", str(out), "-device", "cuda:0",
                      "-batchSize", "8"])
    res = pd.read_csv(out / "gpu_test_predictions.tsv", delimiter="\t")
    emb = np.load(out / "gpu_test_embeddings.npz")["test"]
    assert emb.shape == (len(src), cfg.d_model) and list(res.columns) == ["label", "prediction"]
    om = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    from plantcaduceus_amd import embeddings
    ref = embeddings.extract_embeddings(om, src["sequences"].tolist(), "cpu", 255, CaduceusTokenizer(), batch_size=8)
    assert np.abs(emb - ref).max() / np.abs(ref).max() < 2e-2          # bf16 model (dtype policy) vs fp32 oracle, 2 layers
    want = xgb_predict.XGBJsonClassifier().load_model(str(clf)).predict_proba(emb)[:, 1]
    np.testing.assert_allclose(res["prediction"].to_numpy(), want, rtol=1e-5)


@pytest.mark.gpu
def test_train_cli_on_gpu_from_snapshot(tmp_path, golden_dir, monkeypatch):
    """reference src/train_XGBoost.py main() on the GPU from a snapshot directory: train / valid / test embeddings through the HIP
    path (checked against the oracle on the same checkpoint), the trainer stubbed (xgboost is not installed), the saved JSON
    evaluated natively for the test table, then the -test_only re-run from the caches."""
    import sys
    import types
    from plantcaduceus_amd import xgb_train
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    d = str(tmp_path / "snap")
    cfg, sd = make_synthetic_checkpoint(d, "x", seed=22, stress=False, d_model=128, n_layer=2)
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t")
    rng = np.random.default_rng(1)
    paths = {}
    for name, rows in (("tr", slice(0, 40)), ("va", slice(40, 60)), ("te", slice(60, 90))):
        paths[name] = tmp_path / f"{name}.tsv"
        seqs = src["sequences"].iloc[rows]
        pd.DataFrame({"sequences": seqs, "label": rng.permutation(np.arange(len(seqs)) % 2)}).to_csv(paths[name], sep="\t", index=False)
    stub = types.ModuleType("xgboost")
    stub.XGBClassifier = _StubXGBClassifier
    monkeypatch.setitem(sys.modules, "xgboost", stub)
    out = tmp_path / "out"
    xgb_train.main(["-train", str(paths["tr"]), "-valid", str(paths["va"]), "-test", str(paths["te"]), "-model", d, "-output", str(out),
                    "-device", "cuda:0", "-batchSize", "16"])
    z = np.load(out / "train_valid_embeddings.npz")
    assert z["train"].shape == (40, 128) and z["valid"].shape == (20, 128)
    om = O.OracleForMaskedLM(O.params_from_state_dict(sd, cfg))
    from plantcaduceus_amd import embeddings
    ref = embeddings.extract_embeddings(om, src["sequences"].iloc[40:60].tolist(), "cpu", 255, CaduceusTokenizer(), batch_size=8)
    assert np.abs(z["valid"] - ref).max() / np.abs(ref).max() < 2e-2      # bf16 model (dtype policy) vs fp32 oracle, 2 layers
    clf = xgb_predict.XGBJsonClassifier().load_model(str(out / "seed_42_XGBoost.json"))
    te = np.load(out / "te_embeddings.npz")["test"]
    want = clf.predict_proba(te)[:, 1]
    np.testing.assert_allclose(np.load(out / "seed_42_te_predictions.npz")["predictions"], want, rtol=1e-6)
    assert (out / "seed_42_va_metrics.txt").exists() and (out / "seed_42_te_metrics.txt").exists()
    monkeypatch.setitem(sys.modules, "xgboost", None)                     # the re-run needs no trainer
    os.remove(out / "seed_42_te_predictions.npz")
    xgb_train.main(["-test", str(paths["te"]), "-model", d, "-output", str(out), "-device", "cuda:0", "-test_only"])
    np.testing.assert_allclose(np.load(out / "seed_42_te_predictions.npz")["predictions"], want, rtol=1e-6)


def test_reader_against_a_file_written_by_real_xgboost(tmp_path):
    """where the `xgboost` package exists (not in the build container: this test skips there), a classifier trained and
    saved by xgboost itself must load through XGBJsonClassifier and reproduce xgboost's own predict_proba - the reader is otherwise
    pinned only by hand-built JSON files."""
    xgb = pytest.importorskip("xgboost")
    rng = np.random.default_rng(0)
    X = rng.standard_normal((400, 12)).astype(np.float32)
    X[rng.random(X.shape) < 0.05] = np.nan                      # missing values take the default direction
    y = (np.nan_to_num(X[:, 0]) + 0.5 * np.nan_to_num(X[:, 3]) > 0).astype(int)
    clf = xgb.XGBClassifier(n_estimators=20, max_depth=4, learning_rate=0.3, random_state=1)
    clf.fit(X, y)
    path = str(tmp_path / "real.json")
    clf.save_model(path)
    mine = xgb_predict.XGBJsonClassifier().load_model(path)
    want = clf.predict_proba(X)[:, 1]
    got = xgb_predict.infer_xgboost_model(mine, X)
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=0, atol=2e-6)
