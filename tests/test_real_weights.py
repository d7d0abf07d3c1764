"""tools/real_weights.sh - the one command that pins the oracle's recalled wiring the day a real PlantCaduceus snapshot is at hand
(parity is unpinned offline) - exercised here on a SYNTHETIC snapshot, on CPU: the audit (config keys, tensors,
tied pairs), the census plumbing (example-table + seeded windows through the C oracle and its reference-order emulation; the engine
rows need a GPU and are skipped), the clean skip without a directory, and the loud failure on a config key the forward would ignore."""
import json
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **kw):
    return subprocess.run([os.path.join(ROOT, "tools", "real_weights.sh")] + args, cwd=ROOT, capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="4"), **kw)


def test_skips_cleanly_without_a_snapshot(tmp_path):
    r = _run([])
    assert r.returncode == 0 and "every step SKIPs" in r.stdout and "audit SKIP, known SKIP, census SKIP, e2e SKIP" in r.stdout
    r = _run([str(tmp_path / "nothing_here")])
    assert r.returncode == 0 and "every step SKIPs" in r.stdout


def test_script_on_a_synthetic_snapshot(tmp_path, golden_dir):
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    snap = str(tmp_path / "snap")
    make_synthetic_checkpoint(snap, "x", seed=11, stress=False, d_model=64, n_layer=2)
    tsv = tmp_path / "few.tsv"
    pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:6].to_csv(tsv, sep="\t", index=False)
    out = tmp_path / "out"
    r = _run([snap, "--tsv", str(tsv), "--n-seeded", "5", "--n-emul", "4", "--out", str(out)])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "audit PASS" in r.stdout and "known SKIP" in r.stdout and "census PASS" in r.stdout and "e2e SKIP" in r.stdout
    c = json.load(open(out / "census.json"))
    assert len(c) == 2 and sorted(v["n"] for v in c.values()) == [5, 6]
    for v in c.values():                      # no GPU here: the oracle-vs-oracle row only, and a margin histogram over all windows
        assert list(v["rows"]) == ["reference-order bf16 emulation vs fp32 oracle (no GPU)"] and sum(v["margin_hist"]) == v["n"]
    assert os.path.exists(out / "real_weights.log")


def test_audit_fails_loudly_on_unknown_config_keys_and_untied_weights(tmp_path):
    from safetensors.torch import load_file, save_file
    from plantcaduceus_amd.checkpoint import EMB_KEY, LMHEAD_KEY, audit_snapshot, make_synthetic_checkpoint
    from plantcaduceus_amd.configuration_caduceus import config_from_dict
    snap = str(tmp_path / "snap")
    make_synthetic_checkpoint(snap, "x", seed=12, d_model=64, n_layer=1)
    assert audit_snapshot(snap)["problems"] == []
    cfgp = os.path.join(snap, "config.json")
    raw = json.load(open(cfgp))
    raw2 = dict(raw, use_mamba2=True)
    raw2["ssm_cfg"] = dict(raw["ssm_cfg"], headdim=64)
    with pytest.raises(ValueError, match="use_mamba2"):
        config_from_dict(raw2, strict=True)
    json.dump(raw2, open(cfgp, "w"))
    with pytest.raises(ValueError, match="ssm_cfg.headdim"):
        audit_snapshot(snap)
    r = _run([snap, "--steps", "audit"])
    assert r.returncode == 1 and "audit FAIL" in r.stdout
    json.dump(raw, open(cfgp, "w"))
    # an lm_head stored with other values than the embedding (the engine reads the embedding), a missing tensor, a stray one
    sd = load_file(os.path.join(snap, "model.safetensors"))
    sd[LMHEAD_KEY] = sd[EMB_KEY] + 1.0
    sd["caduceus.backbone.layers.0.mixer.submodule.mamba_fwd.extra.weight"] = sd[EMB_KEY][:1].clone()
    del sd["caduceus.backbone.layers.0.norm.weight"]
    save_file(sd, os.path.join(snap, "model.safetensors"))
    rep = audit_snapshot(snap, strict=False)
    text = " | ".join(rep["problems"])
    assert "missing tensors" in text and "does not read" in text
    assert np.isfinite(1.0)


@pytest.mark.gpu
def test_census_engine_rows_on_a_synthetic_snapshot(tmp_path, golden_dir):
    """the GPU half of step (ii): on a ROCm device the census adds the engine rows (bf16 in its three operation orders, fp32,
    fp32 + f32_gemm_split) against the fp32 oracle; on a (small, well-conditioned) synthetic snapshot the fp32 engines make
    exactly the oracle's calls."""
    import importlib.util
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    snap = str(tmp_path / "snap")
    make_synthetic_checkpoint(snap, "x", seed=11, stress=False, d_model=128, n_layer=2)
    tsv = tmp_path / "few.tsv"
    pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:9].to_csv(tsv, sep="\t", index=False)
    spec = importlib.util.spec_from_file_location("argmax_census", os.path.join(ROOT, "tools", "argmax_census.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    res = mod.census_on_snapshot(snap, tsv=str(tsv), n_seeded=16, n_emul=8, out=lines.append)
    print("\n".join(lines))
    assert len(res) == 2
    for v in res.values():
        rows = v["rows"]
        assert "HIP fp32 vs fp32 oracle" in rows and "HIP fp32 + f32_gemm_split vs fp32 oracle" in rows
        assert rows["HIP fp32 vs fp32 oracle"]["flips"] == 0 and rows["HIP fp32 + f32_gemm_split vs fp32 oracle"]["flips"] == 0
        assert rows["HIP fp32 + f32_gemm_split vs fp32 oracle"]["max_dp"] < 1e-4
        assert sum(1 for k in rows if k.startswith("HIP bf16")) >= 3
        assert all(r["max_dp"] < 5e-2 for r in rows.values())
