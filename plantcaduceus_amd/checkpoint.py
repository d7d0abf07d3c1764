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


def make_synthetic_checkpoint(path: str, size: str = "l20", seed: int = 1234, stress: bool = True, **overrides):
    cfg = make_config(size, **overrides)
    sd = synthetic_state_dict(cfg, seed=seed, stress=stress)
    save_checkpoint(path, cfg, sd)
    return cfg, sd
