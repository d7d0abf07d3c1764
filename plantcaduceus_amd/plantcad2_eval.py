"""Probability primitives of the reference's PlantCAD2 evaluation CLI (`src/zero-shot-eval.py`), on the same
`model(input_ids=...).logits` boundary as the zero-shot path:

  masked_probs    `SingleMaskDataset` / `MultiMaskDataset` + `_masked_probs`  (:75-140): mask one or several positions of
                  every sequence, softmax over a,c,g,t of the logits AT the masked positions -> [N * n_mask, 4], rows in
                  (sequence, ascending position) order (the order of the reference's `torch.masked_select`)
  unmasked_probs  `_unmasked_probs` (:143-178): per-position probabilities of the un-masked sequence -> [N, L, 4]

Only the head rows that are read are evaluated (`positions=`); tokenisation is the vectorised LUT.  The task drivers
built on these in the reference (SV boundary LLR, AUROC tables, HF `datasets` loading) are CPU-side bookkeeping and
are not reproduced.
"""
from __future__ import annotations

from typing import Sequence, Union

import numpy as np
import torch

from .zero_shot import tokenize_masked

NUCLEOTIDES_LOWER = ("a", "c", "g", "t")


def masked_probs(model, tokenizer, sequences: Sequence[str], mask_idx: Union[int, Sequence[int]], device,
                 batch_size: int = 128) -> np.ndarray:
    idx = sorted({int(mask_idx)} if isinstance(mask_idx, (int, np.integer)) else {int(i) for i in mask_idx})
    cols = [tokenizer.get_vocab()[n] for n in NUCLEOTIDES_LOWER]
    ids_all = torch.from_numpy(tokenize_masked(list(sequences), tokenizer, None).astype(np.int64))
    if ids_all.shape[0] and ids_all.shape[1] <= max(idx):
        raise ValueError(f"mask index {max(idx)} out of range for sequence length {ids_all.shape[1]}")
    ids_all[:, idx] = tokenizer.mask_token_id
    fast = bool(getattr(model, "supports_positions", False)) and len(idx) <= 16
    out = []
    with torch.inference_mode():
        for b0 in range(0, ids_all.shape[0], batch_size):
            cur = ids_all[b0:b0 + batch_size].to(device)
            lg = model(input_ids=cur, positions=idx).logits if fast else model(input_ids=cur).logits[:, idx, :]
            out.append(torch.softmax(lg[..., cols].float(), dim=-1).reshape(-1, 4).cpu().numpy())
    return np.vstack(out) if out else np.zeros((0, 4), dtype=np.float32)


def unmasked_probs(sequences: Sequence[str], tokenizer, model, device, batch_size: int = 32) -> np.ndarray:
    cols = [tokenizer.get_vocab()[n] for n in NUCLEOTIDES_LOWER]
    seqs = [str(s) for s in sequences]
    if len({len(s) for s in seqs}) > 1:
        raise ValueError("All sequences must have same length")
    ids_all = torch.from_numpy(tokenize_masked(seqs, tokenizer, None).astype(np.int64))
    out = np.zeros((len(seqs), ids_all.shape[1] if len(seqs) else 0, 4), dtype=np.float32)
    with torch.inference_mode():
        for b0 in range(0, len(seqs), batch_size):
            lg = model(input_ids=ids_all[b0:b0 + batch_size].to(device)).logits[..., cols]
            out[b0:b0 + batch_size] = torch.softmax(lg.float(), dim=-1).cpu().numpy()
    return out
