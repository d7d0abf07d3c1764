"""Checkpoints: reference state-dict key names, a deterministic synthetic generator and a loader.

There are no PlantCaduceus weights offline, so benchmarks and tests use synthetic checkpoints written
as `config.json` + `model.safetensors` with the *reference key names* (module tree printed at reference
`notebooks/examples.ipynb:61-100`) so that a real HF snapshot directory loads through
exactly the same path.  Init scheme mirrors `mamba_ssm.Mamba.__init__` / torch `nn.Linear`/`nn.Conv1d`
defaults (restated at transformers `models/mamba/modeling_mamba.py:337-356`), plus a "stress" variant
(perturbed A_log, non-unit norm weights / D) that makes direction / channel / strand bugs visible.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict

import numpy as np
import torch

from .configuration_caduceus import CaduceusConfig, PLANTCADUCEUS_SIZES

BACKBONE = "caduceus.backbone."
EMB_KEY = BACKBONE + "embeddings.word_embeddings.embedding.weight"
NORMF_KEY = BACKBONE + "norm_f.weight"
LMHEAD_KEY = "lm_head.lm_head.weight"


def layer_keys(i: int, direction: str) -> Dict[str, str]:
    mp = f"{BACKBONE}layers.{i}.mixer.submodule.mamba_{direction}."
    return dict(
        in_proj=mp + "in_proj.weight", conv_w=mp + "conv1d.weight", conv_b=mp + "conv1d.bias",
        x_proj=mp + "x_proj.weight", dt_w=mp + "dt_proj.weight", dt_b=mp + "dt_proj.bias",
        A_log=mp + "A_log", D=mp + "D", out_proj=mp + "out_proj.weight",
    )


def norm_key(i: int) -> str:
    return f"{BACKBONE}layers.{i}.norm.weight"


def make_config(size: str = "l20", **overrides) -> CaduceusConfig:
    kw = dict(PLANTCADUCEUS_SIZES[size]) if size in PLANTCADUCEUS_SIZES else {}
    kw.update(overrides)
    kw.setdefault("ssm_cfg", dict(d_state=16, d_conv=4, expand=2, dt_rank="auto", bias=False, conv_bias=True))
    return CaduceusConfig(vocab_size=8, **kw)


def synthetic_state_dict(config: CaduceusConfig, seed: int = 1234, stress: bool = True) -> Dict[str, torch.Tensor]:
    rng = np.random.default_rng(seed)
    D, E, N, R, W, V = (config.d_model, config.d_inner, config.d_state, config.dt_rank, config.d_conv,
                        config.padded_vocab_size)

    def U(shape, bound):
        return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))

    sd: Dict[str, torch.Tensor] = {}
    emb_scale = 0.5 if stress else 0.02
    sd[EMB_KEY] = torch.from_numpy((rng.standard_normal((V, D)) * emb_scale).astype(np.float32))
    for i in range(config.n_layer):
        sd[norm_key(i)] = (torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32)) if stress
                           else torch.ones(D))
        in_proj = U((2 * E, D), D ** -0.5)
        out_proj = U((D, E), E ** -0.5) / math.sqrt(config.n_layer)
        for d in ("fwd", "rev"):
            k = layer_keys(i, d)
            sd[k["in_proj"]] = in_proj          # tied between directions
            sd[k["out_proj"]] = out_proj        # tied
            sd[k["conv_w"]] = U((E, 1, W), W ** -0.5)
            sd[k["conv_b"]] = U((E,), W ** -0.5)
            sd[k["x_proj"]] = U((R + 2 * N, E), E ** -0.5)
            sd[k["dt_w"]] = U((E, R), R ** -0.5)
            dt = np.exp(rng.uniform(size=E) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3)).clip(min=1e-4)
            sd[k["dt_b"]] = torch.from_numpy((dt + np.log(-np.expm1(-dt))).astype(np.float32))
            A_log = np.log(np.tile(np.arange(1, N + 1, dtype=np.float64), (E, 1)))
            if stress:
                A_log = A_log + rng.normal(0, 0.3, size=A_log.shape)
            sd[k["A_log"]] = torch.from_numpy(A_log.astype(np.float32))
            sd[k["D"]] = (torch.from_numpy(rng.uniform(0.5, 1.5, E).astype(np.float32)) if stress
                          else torch.ones(E))
    sd[NORMF_KEY] = (torch.from_numpy(rng.uniform(0.5, 1.5, D).astype(np.float32)) if stress
                     else torch.ones(D))
    sd[LMHEAD_KEY] = sd[EMB_KEY]                # tied
    return sd


def harsh_state_dict(config: CaduceusConfig, seed: int = 21, proj_scale: float = 1.0, dt_scale: float = 256.0) -> Dict[str, torch.Tensor]:
    """The `stress` checkpoint (distinct fwd/rev parameters, perturbed A_log / D / norm weights) with every dt_proj weight scaled
    by `dt_scale` (and every in_proj / x_proj weight by `proj_scale`): time steps large enough that ~12 % of the (t, channel)
    elements take softplus's pass-through branch (delta + bias > 20) — the regime the benign benchmark checkpoint never reaches
    (tools/argmax_census.py, tests/test_gpu_fulldepth.py).  The defaults keep the 32-layer stack well conditioned: two fp32 CPU
    restatements that differ only in summation order (oracle/c with and without BLAS) agree to 6e-8 on it, so north_star's 1e-4
    is a meaningful bar.  Scaling the projections as well (proj_scale 4, dt_scale 16: |x| >> 1 through the stack) makes the
    network amplify rounding noise ~1e5-fold — those two restatements then differ by 4e-2 — so on that variant a comparison can
    only be read against that noise floor (profiles/r03_argmax_census.txt prints both)."""
    sd = synthetic_state_dict(config, seed=seed, stress=True)
    for k in list(sd):
        if k.endswith("in_proj.weight") or k.endswith("x_proj.weight"):
            sd[k] = sd[k] * proj_scale
        elif k.endswith("dt_proj.weight"):
            sd[k] = sd[k] * dt_scale
    return sd


def save_checkpoint(path: str, config: CaduceusConfig, sd: Dict[str, torch.Tensor], with_tokenizer: bool = True):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = config.to_dict()
    cfg["architectures"] = ["CaduceusForMaskedLM"]
    cfg["model_type"] = "caduceus"
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2, default=str)
    # safetensors refuses shared storage: drop the tied duplicates exactly as a real snapshot does
    out = {}
    for k, v in sd.items():
        if k == LMHEAD_KEY:
            continue
        if ".mamba_rev.in_proj." in k or ".mamba_rev.out_proj." in k:
            continue
        out[k] = v.contiguous().clone()
    save_file(out, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    if with_tokenizer:
        from .tokenization_caduceus import CaduceusTokenizer
        CaduceusTokenizer().save_pretrained(path)


def resolve_snapshot(name_or_path: str, **hub_kwargs) -> str:
    """A local snapshot directory as is; anything else is taken as a HF-hub repo id (the reference's README and CLI pass
    `-model kuleshov-group/PlantCaduceus_l32`) and resolved with `huggingface_hub.snapshot_download`, which honours
    HF_HUB_OFFLINE / `local_files_only` (served from the local HF cache when there is no network)."""
    path = str(name_or_path)
    if os.path.isdir(path):
        return path
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(repo_id=path, allow_patterns=["*.json", "*.safetensors", "*.bin", "*.txt"], **hub_kwargs)
    except Exception as e:   # offline and not cached, bad id, ...
        raise OSError(f"{path} is neither a local snapshot directory nor a HF-hub repo that can be resolved here ({e}); "
                      "download kuleshov-group/PlantCaduceus_l* and pass its directory, or populate the HF cache") from e


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Read a snapshot directory — model.safetensors, sharded model.safetensors.index.json / pytorch_model.bin.index.json,
    or pytorch_model.bin — and restore the tied keys."""
    def _one(fn):
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(os.path.join(path, fn))
        return torch.load(os.path.join(path, fn), map_location="cpu", weights_only=True)

    sd: Dict[str, torch.Tensor] = {}
    for single, index in (("model.safetensors", "model.safetensors.index.json"),
                          ("pytorch_model.bin", "pytorch_model.bin.index.json")):
        if os.path.exists(os.path.join(path, single)):
            sd = dict(_one(single))
            break
        if os.path.exists(os.path.join(path, index)):
            with open(os.path.join(path, index)) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
            for fn in shards:
                sd.update(_one(fn))
            break
    else:
        raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin (or their .index.json shards)")
    if LMHEAD_KEY not in sd:
        sd[LMHEAD_KEY] = sd[EMB_KEY]
    for k in list(sd.keys()):
        if ".mamba_fwd.in_proj." in k or ".mamba_fwd.out_proj." in k:
            rk = k.replace(".mamba_fwd.", ".mamba_rev.")
            sd.setdefault(rk, sd[k])
    return sd


def expected_keys(config: CaduceusConfig) -> Dict[str, tuple]:
    """name -> shape of every tensor `CaduceusForMaskedLM` of this configuration holds (module tree of reference
    notebooks/examples.ipynb:61-100; SURVEY.md §8a "State-dict key names")."""
    D, E, N, R, W, V = (config.d_model, config.d_inner, config.d_state, config.dt_rank, config.d_conv, config.padded_vocab_size)
    exp = {EMB_KEY: (V, D), NORMF_KEY: (D,), LMHEAD_KEY: (V, D)}
    for i in range(config.n_layer):
        exp[norm_key(i)] = (D,)
        for d in ("fwd", "rev"):
            k = layer_keys(i, d)
            exp.update({k["in_proj"]: (2 * E, D), k["conv_w"]: (E, 1, W), k["conv_b"]: (E,), k["x_proj"]: (R + 2 * N, E),
                        k["dt_w"]: (E, R), k["dt_b"]: (E,), k["A_log"]: (E, N), k["D"]: (E,), k["out_proj"]: (D, E)})
    return exp


def audit_snapshot(path: str, strict: bool = True) -> dict:
    """Everything a hub snapshot directory must satisfy for this implementation to reproduce the reference on it, checked without
    a GPU: config.json keys (configuration_caduceus.audit_config_dict: unknown keys fail), the configuration is one the engine
    implements, every tensor of the module tree is present with the right shape, tensors the forward does not read are listed
    (only `...complement_map` buffers are expected among them), the tied pairs really are tied (lm_head = embedding; mamba_rev
    in_proj / out_proj = mamba_fwd's) where the file stores both.  -> report dict; problems raise ValueError when strict."""
    from .configuration_caduceus import audit_config_dict, config_from_dict
    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    rep = {"config_keys": audit_config_dict(raw), "problems": []}
    cfg = config_from_dict(raw, strict=False)
    bad_keys = rep["config_keys"]["unknown"] + ["ssm_cfg." + k for k in rep["config_keys"]["ssm_unknown"]]
    if bad_keys:
        rep["problems"].append("config.json keys this implementation would ignore: %s" % bad_keys)
    try:
        cfg.check_supported()
    except ValueError as e:
        rep["problems"].append(str(e))
    rep["geometry"] = dict(d_model=cfg.d_model, n_layer=cfg.n_layer, d_inner=cfg.d_inner, d_state=cfg.d_state, dt_rank=cfg.dt_rank,
                           vocab=cfg.padded_vocab_size, norm_epsilon=cfg.norm_epsilon, residual_in_fp32=bool(cfg.residual_in_fp32))
    # the raw file contents (before load_state_dict restores the tied keys), so that stored duplicates can be compared
    sd = load_state_dict(path)
    exp = expected_keys(cfg)
    missing = [k for k in exp if k not in sd]
    wrong = [(k, tuple(sd[k].shape), exp[k]) for k in exp if k in sd and tuple(sd[k].shape) != tuple(exp[k])]
    extra = [k for k in sd if k not in exp]
    unexpected = [k for k in extra if not k.endswith("complement_map")]
    if missing:
        rep["problems"].append("missing tensors: %s%s" % (missing[:4], " ..." if len(missing) > 4 else ""))
    if wrong:
        rep["problems"].append("shape mismatches (name, file, expected): %s" % wrong[:4])
    if unexpected:
        rep["problems"].append("tensors the forward does not read: %s%s" % (unexpected[:4], " ..." if len(unexpected) > 4 else ""))
    untied = []
    if not missing and not wrong:
        if not torch.equal(sd[LMHEAD_KEY], sd[EMB_KEY]):
            untied.append(LMHEAD_KEY)
        for i in range(cfg.n_layer):
            f, r = layer_keys(i, "fwd"), layer_keys(i, "rev")
            for nm in ("in_proj", "out_proj"):
                if not torch.equal(sd[f[nm]], sd[r[nm]]):
                    untied.append(r[nm])
        cm = [k for k in extra if k.endswith("complement_map")]
        for k in cm:
            got = [int(v) for v in sd[k].flatten().tolist()]
            if got != cfg.complement_list()[:len(got)]:
                rep["problems"].append("%s = %s differs from config.complement_map -> %s" % (k, got, cfg.complement_list()))
    if untied:
        rep["problems"].append("tied tensors stored with different values (the engine reads the first of each pair): %s" % untied[:4])
    rep["tensors"] = dict(expected=len(exp), in_file=len(sd), extra=extra[:8], dtypes=sorted({str(v.dtype) for v in sd.values()}))
    if strict and rep["problems"]:
        raise ValueError("snapshot %s does not pass the audit:\n  - " % path + "\n  - ".join(rep["problems"]))
    return rep


def make_synthetic_checkpoint(path: str, size: str = "l20", seed: int = 1234, stress: bool = True, **overrides):
    cfg = make_config(size, **overrides)
    sd = synthetic_state_dict(cfg, seed=seed, stress=stress)
    save_checkpoint(path, cfg, sd)
    return cfg, sd
