"""Character-level DNA tokenizer with the surface the reference's hot-path scripts use.

Call sites reproduced: `tokenizer.encode_plus(seq, return_tensors="pt", return_attention_mask=False,
return_token_type_ids=False)['input_ids']` -> LongTensor [1, L] with NO special tokens added
(reference `src/zero_shot_score.py:51-57`, `src/train_XGBoost.py:39-45`), `tokenizer.get_vocab()['a'|'c'|'g'|'t']`
(`src/zero_shot_score.py:118`), `tokenizer.mask_token_id` (`:58`).  transformers>=5 dropped
`encode_plus`; it is provided here as a shim over `__call__`.

Vocabulary `[PAD]0 [MASK]1 [UNK]2 a3 c4 g5 t6` (+ one pad row, vocab padded to 8: reference
`pretrain/llmlib/architectures/models/mamba/caduceus.py:124-125`); input is lower-cased; any other
character (N, IUPAC codes) maps to [UNK].  If the model directory carries its own `vocab.json`, that
vocabulary wins.

`encode_batch` is the engine's host fast path: a vectorised byte->id LUT over a whole batch, replacing
the reference's per-sequence Python tokenisation in the main process.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from transformers import PreTrainedTokenizer

DEFAULT_VOCAB = {"[PAD]": 0, "[MASK]": 1, "[UNK]": 2, "a": 3, "c": 4, "g": 5, "t": 6}
VOCAB_FILE = "vocab.json"


class CaduceusTokenizer(PreTrainedTokenizer):
    vocab_files_names = {"vocab_file": VOCAB_FILE}
    model_input_names = ["input_ids"]

    def __init__(self, vocab_file: Optional[str] = None, vocab: Optional[Dict[str, int]] = None,
                 model_max_length: int = 512, **kwargs):
        if vocab is None and vocab_file is not None and os.path.exists(vocab_file):
            with open(vocab_file) as f:
                vocab = json.load(f)
        self._vocab = dict(vocab) if vocab else dict(DEFAULT_VOCAB)
        self._ids_to_tokens = {v: k for k, v in self._vocab.items()}
        kwargs.setdefault("pad_token", "[PAD]")
        kwargs.setdefault("mask_token", "[MASK]")
        kwargs.setdefault("unk_token", "[UNK]")
        kwargs.pop("add_prefix_space", None)
        super().__init__(model_max_length=model_max_length, **kwargs)
        self._lut = self._build_lut()

    # ---- HF slow-tokenizer protocol ---------------------------------------------------------
    @property
    def vocab_size(self) -> int:
        return len(self._vocab)

    def get_vocab(self) -> Dict[str, int]:
        return dict(self._vocab)

    def _tokenize(self, text: str, **kwargs) -> List[str]:
        return list(text.lower())

    def _convert_token_to_id(self, token: str) -> int:
        return self._vocab.get(token, self._vocab["[UNK]"])

    def _convert_id_to_token(self, index: int) -> str:
        return self._ids_to_tokens.get(int(index), "[UNK]")

    def convert_tokens_to_string(self, tokens: List[str]) -> str:
        return "".join(tokens)

    def build_inputs_with_special_tokens(self, token_ids_0, token_ids_1=None):
        return list(token_ids_0) if token_ids_1 is None else list(token_ids_0) + list(token_ids_1)

    def num_special_tokens_to_add(self, pair: bool = False) -> int:
        return 0

    def save_vocabulary(self, save_directory: str, filename_prefix: Optional[str] = None):
        path = os.path.join(save_directory, (filename_prefix + "-" if filename_prefix else "") + VOCAB_FILE)
        with open(path, "w") as f:
            json.dump(self._vocab, f)
        return (path,)

    # ---- surface used by the reference scripts ------------------------------------------------
    def _encode_ids(self, text: str) -> np.ndarray:
        return self._lut[np.frombuffer(text.encode("latin-1", "replace"), dtype=np.uint8)]

    def __call__(self, text, return_tensors=None, add_special_tokens=False, **kwargs):
        single = isinstance(text, str)
        seqs = [text] if single else list(text)
        ids = [self._encode_ids(s) for s in seqs]
        if return_tensors == "pt":
            if len({len(i) for i in ids}) != 1:
                raise ValueError("sequences of unequal length cannot be stacked (no padding on this path)")
            return {"input_ids": torch.from_numpy(np.stack(ids).astype(np.int64))}
        if return_tensors == "np":
            return {"input_ids": np.stack(ids).astype(np.int64)}
        out = [i.tolist() for i in ids]
        return {"input_ids": out[0] if single else out}

    def encode_plus(self, text, return_tensors=None, **kwargs):
        """Shim for the pre-v5 API the reference still calls (src/zero_shot_score.py:51)."""
        return self.__call__(text, return_tensors=return_tensors)

    def encode(self, text, **kwargs) -> List[int]:
        return self._encode_ids(text).tolist()

    # ---- vectorised host path ----------------------------------------------------------------
    def _build_lut(self) -> np.ndarray:
        lut = np.full(256, self._vocab["[UNK]"], dtype=np.int32)
        for tok, idx in self._vocab.items():
            if len(tok) == 1:
                lut[ord(tok.lower())] = idx
                lut[ord(tok.upper())] = idx
        return lut

    def encode_batch(self, sequences: Sequence[str], mask_index: Optional[int] = None) -> np.ndarray:
        """[N] equal-length strings -> int32 [N, L]; optionally overwrite column `mask_index` with
        [MASK] (reference src/zero_shot_score.py:58)."""
        if len(sequences) == 0:
            return np.zeros((0, 0), dtype=np.int32)
        L = len(sequences[0])
        buf = "".join(sequences).encode("latin-1", "replace")
        if len(buf) != L * len(sequences):
            raise ValueError("all sequences must have equal length")
        ids = self._lut[np.frombuffer(buf, dtype=np.uint8)].reshape(len(sequences), L)
        if mask_index is not None:
            ids = ids.copy()
            ids[:, mask_index] = self.mask_token_id
        return ids
