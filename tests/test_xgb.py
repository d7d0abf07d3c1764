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
