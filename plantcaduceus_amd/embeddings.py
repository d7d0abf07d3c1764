"""Embedding extraction — host side of the hot path, mirroring reference `src/train_XGBoost.py`
(`extract_embeddings` :96-114, `SequenceDataset` :29-49, `load_data` :51-54; `predict_XGBoost.py`
reaches the same function through `from train_XGBoost import *`).

  extract_embeddings(model, sequences, device, tokenIdx, tokenizer, batch_size)
      forward with output_hidden_states=True, hidden_states[-1][:, tokenIdx, :] -> fp32,
      forward half + channel-reversed rc half, / 2                                   -> fp32 [N, d_model]

Host-side differences only: whole-list vectorised tokenisation (no masking on this path), the model is
asked for position tokenIdx only (hidden [B,1,2D] instead of 33 x [B,512,2D]), the averaging runs on the
device, and under torch.distributed the window list is block-sharded and the [N, D] result reassembled with
one all-gather per call (= one `-chunk_size` chunk of the reference's -save_memory mode, :175-190).
The XGBoost fit/predict and the ROC/PR plots of the reference script are CPU tree-model code and are out of
scope; `save_embedding_cache` keeps the reference's .npz keys so they run unchanged.
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from . import sharding
from .zero_shot import _window_len, check_model_inputs, effective_batch, iter_device_batches, local_ids


def load_data(filepath):
    import pandas as pd
    logging.info(f"Loading data from {filepath}")
    data = pd.read_csv(filepath, delimiter="\t")
    return data["sequences"].tolist(), data["label"].tolist()


def extract_embeddings(model, sequences, device, tokenIdx: int, tokenizer=None, batch_size: int = 128,
                       batch_explicit: bool = False) -> np.ndarray:
    logging.info("Extracting embeddings")
    n_total = len(sequences)
    rank, ws = sharding.world()
    start, stop, per = sharding.shard_bounds(n_total, rank, ws)            # only this rank's block is tokenised
    fast = bool(getattr(model, "supports_positions", False))
    if fast and n_total:
        batch_size = effective_batch(model, batch_size, _window_len(sequences), batch_explicit)
    model.eval()
    outs = []
    with torch.inference_mode():
        for cur in iter_device_batches(sequences, start, stop, per if ws > 1 else 0, batch_size, tokenizer, None, device):
            if fast:
                e = model(input_ids=cur, output_hidden_states=True, positions=[tokenIdx]).hidden_states[-1][:, 0, :]
            else:
                e = model(input_ids=cur, output_hidden_states=True).hidden_states[-1][:, tokenIdx, :]
            e = e.to(torch.float32)
            hsz = e.shape[-1] // 2
            outs.append((e[:, :hsz] + torch.flip(e[:, hsz:], dims=[-1])) / 2)
        if outs:
            emb = torch.cat(outs, dim=0)
        else:
            d = getattr(getattr(model, "config", None), "d_model", 0)
            emb = torch.empty((0, d), dtype=torch.float32, device=device)
        emb = sharding.all_gather_rows(emb, n_total)
    out = emb.cpu().numpy()
    check_model_inputs(model)
    return out


def save_embedding_cache(path: str, compresslevel: int = 1, **arrays):
    """The reference's cache file (`np.savez_compressed` with keys `train`, `valid`, `test`: src/train_XGBoost.py:186,199,221),
    readable by `np.load` exactly as the reference reads it, written with a chosen deflate level (0 = stored).  fp32 embeddings
    hardly compress (7 %) and deflate runs at ~20 MB/s on one core at any level (2.3 s at level 1, 2.8 s at numpy's default, for
    the 51 MB of one rank's share of a chunk; 6 s on the GPU box's slower cores, profiles/r03_e2e_embed.json), so callers that
    extract chunk after chunk write the file on a worker thread while the next chunk runs (xgb_predict.py)."""
    import zipfile
    bad = set(arrays) - {"train", "valid", "test"}
    if bad:
        raise ValueError(f"unexpected cache keys {sorted(bad)}")
    if not path.endswith(".npz"):
        path = path + ".npz"                       # np.savez_compressed appends the suffix the same way
    comp = zipfile.ZIP_DEFLATED if compresslevel > 0 else zipfile.ZIP_STORED
    # written beside the final name and renamed once the archive is closed: a run that dies mid-write (the writer may be a
    # background thread that lives for a whole chunk's compute time) leaves a .tmp file, never a truncated cache that the next
    # run would find with os.path.exists() and fail to read
    tmp = path + ".tmp"
    try:
        with zipfile.ZipFile(tmp, "w", compression=comp, compresslevel=compresslevel if compresslevel > 0 else None, allowZip64=True) as zf:
            for key, arr in arrays.items():
                with zf.open(key + ".npy", "w", force_zip64=True) as f:
                    np.lib.format.write_array(f, np.asanyarray(arr), allow_pickle=False)
        os.replace(tmp, path)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise
