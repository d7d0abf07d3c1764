"""CaduceusConfig — the `config.json` schema of the `kuleshov-group/PlantCaduceus_l*` snapshots.

The reference loads this class through `trust_remote_code=True`
(reference `src/zero_shot_score.py:91`, `src/train_XGBoost.py:87`); the field set is the one the
reference's own pre-training wrapper passes to `AutoConfig`
(reference `pretrain/llmlib/architectures/models/mamba/caduceus.py:100-128`: `complement_map`,
vocab padded to a multiple of 8) and the one implied by the module tree printed at
reference `notebooks/examples.ipynb:61-100`.  Model hyper-parameters are always read from the
snapshot's `config.json`, never hard-coded.
"""
from __future__ import annotations

import math
from typing import Optional

from transformers import PretrainedConfig

# id → id of the complementary base; vocabulary `[PAD]0 [MASK]1 [UNK]2 a3 c4 g5 t6 (pad row)7`
DEFAULT_COMPLEMENT_MAP = {0: 0, 1: 1, 2: 2, 3: 6, 4: 5, 5: 4, 6: 3, 7: 7}


class CaduceusConfig(PretrainedConfig):
    model_type = "caduceus"

    def __init__(
        self,
        d_model: int = 384,
        n_layer: int = 20,
        vocab_size: int = 8,
        ssm_cfg: Optional[dict] = None,
        rms_norm: bool = True,
        residual_in_fp32: bool = True,
        fused_add_norm: bool = True,
        pad_vocab_size_multiple: int = 8,
        norm_epsilon: float = 1e-5,
        initializer_cfg: Optional[dict] = None,
        bidirectional: bool = True,
        bidirectional_strategy: str = "add",
        bidirectional_weight_tie: bool = True,
        rcps: bool = True,
        complement_map: Optional[dict] = None,
        **kwargs,
    ):
        super().__init__(**kwargs)
        self.d_model = d_model
        self.n_layer = n_layer
        self.vocab_size = vocab_size
        self.ssm_cfg = dict(ssm_cfg) if ssm_cfg else {}
        self.rms_norm = rms_norm
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.pad_vocab_size_multiple = pad_vocab_size_multiple
        self.norm_epsilon = norm_epsilon
        self.initializer_cfg = initializer_cfg
        self.bidirectional = bidirectional
        self.bidirectional_strategy = bidirectional_strategy
        self.bidirectional_weight_tie = bidirectional_weight_tie
        self.rcps = rcps
        if complement_map is None:
            complement_map = DEFAULT_COMPLEMENT_MAP
        # json round-trips dict keys as strings
        self.complement_map = {int(k): int(v) for k, v in dict(complement_map).items()}

    # ---- derived Mamba dimensions (mamba_ssm.Mamba defaults) -------------------------------
    @property
    def padded_vocab_size(self) -> int:
        m = self.pad_vocab_size_multiple
        return ((self.vocab_size + m - 1) // m) * m if m else self.vocab_size

    @property
    def d_state(self) -> int:
        return int(self.ssm_cfg.get("d_state", 16))

    @property
    def d_conv(self) -> int:
        return int(self.ssm_cfg.get("d_conv", 4))

    @property
    def expand(self) -> int:
        return int(self.ssm_cfg.get("expand", 2))

    @property
    def d_inner(self) -> int:
        return self.expand * self.d_model

    @property
    def dt_rank(self) -> int:
        r = self.ssm_cfg.get("dt_rank", "auto")
        return math.ceil(self.d_model / 16) if r == "auto" else int(r)

    def complement_list(self) -> list:
        v = self.padded_vocab_size
        out = list(range(v))
        for k, c in self.complement_map.items():
            if k < v:
                out[k] = c
        return out

    def check_supported(self) -> None:
        """The MI355X engine implements exactly the PlantCaduceus family's configuration."""
        bad = []
        if not self.rcps:
            bad.append("rcps=False")
        if not self.bidirectional or self.bidirectional_strategy != "add":
            bad.append("bidirectional must be True with strategy 'add'")
        if not self.bidirectional_weight_tie:
            bad.append("bidirectional_weight_tie=False")
        if not self.rms_norm:
            bad.append("rms_norm=False")
        if self.d_conv != 4:
            bad.append(f"d_conv={self.d_conv} (only 4)")
        if self.d_state != 16:
            bad.append(f"d_state={self.d_state} (only 16)")
        if self.ssm_cfg.get("bias", False):
            bad.append("ssm_cfg.bias=True")
        if not self.ssm_cfg.get("conv_bias", True):
            bad.append("ssm_cfg.conv_bias=False")
        if self.d_model % 64:
            bad.append(f"d_model={self.d_model} not a multiple of 64")
        if self.padded_vocab_size != 8:
            bad.append(f"padded vocab {self.padded_vocab_size} != 8")
        if bad:
            raise ValueError("unsupported Caduceus configuration for the MI355X engine: " + "; ".join(bad))


# ---- config.json audit (tools/real_weights.sh; CaduceusForMaskedLM.from_pretrained warns with it) ----------------------------------
# Keys of a hub snapshot's config.json, by what this implementation does with them.  PretrainedConfig accepts ANY keyword and
# stores it as an attribute, so a key this forward would need to honour but does not know would be ignored silently: the audit
# names every key and `config_from_dict(strict=True)` refuses the unknown ones.
_CONSUMED_KEYS = ("d_model", "n_layer", "vocab_size", "ssm_cfg", "rms_norm", "residual_in_fp32", "fused_add_norm",
                  "pad_vocab_size_multiple", "norm_epsilon", "initializer_cfg", "bidirectional", "bidirectional_strategy",
                  "bidirectional_weight_tie", "rcps", "complement_map")
# written by save_pretrained / the hub machinery; no effect on the arithmetic of the forward
_BOOKKEEPING_KEYS = ("architectures", "auto_map", "model_type", "torch_dtype", "dtype", "transformers_version", "_name_or_path",
                     "_commit_hash", "_attn_implementation_autoset", "tokenizer_class", "name_or_path")
# mamba_ssm.Mamba keyword arguments (mamba-ssm 2.2.2): the first group shapes the forward and is read here; the second only
# initialises parameters (training) or picks among numerically equivalent code paths
# attributes transformers 4.x's PretrainedConfig (the reference pins 4.40.0, env/requirements.txt:8) may serialise beyond the installed version's
_HF_GENERIC_4X = ("return_dict", "output_hidden_states", "output_attentions", "torchscript", "use_bfloat16", "tf_legacy_loss",
                  "pruned_heads", "tie_word_embeddings", "chunk_size_feed_forward", "is_encoder_decoder", "is_decoder",
                  "cross_attention_hidden_size", "add_cross_attention", "tie_encoder_decoder", "max_length", "min_length",
                  "do_sample", "early_stopping", "num_beams", "num_beam_groups", "diversity_penalty", "temperature", "top_k",
                  "top_p", "typical_p", "repetition_penalty", "length_penalty", "no_repeat_ngram_size",
                  "encoder_no_repeat_ngram_size", "bad_words_ids", "num_return_sequences", "output_scores",
                  "return_dict_in_generate", "forced_bos_token_id", "forced_eos_token_id", "remove_invalid_values",
                  "exponential_decay_length_penalty", "suppress_tokens", "begin_suppress_tokens", "finetuning_task", "id2label",
                  "label2id", "prefix", "bos_token_id", "pad_token_id", "eos_token_id", "sep_token_id",
                  "decoder_start_token_id", "task_specific_params", "problem_type")
_SSM_FORWARD_KEYS = ("d_state", "d_conv", "expand", "dt_rank", "conv_bias", "bias")
_SSM_INIT_KEYS = ("dt_min", "dt_max", "dt_init", "dt_scale", "dt_init_floor", "use_fast_path", "layer_idx", "device", "dtype")


def audit_config_dict(raw: dict) -> dict:
    """-> {"consumed": [...], "bookkeeping": [...], "hf_generic": [...], "unknown": [...], "ssm_unknown": [...]} for the keys of a
    config.json.  hf_generic = attributes every PretrainedConfig serialises (return_dict, id2label, ...), which the masked-LM forward
    does not read."""
    generic = set(PretrainedConfig().to_dict().keys()) | set(_HF_GENERIC_4X)
    out = {"consumed": [], "bookkeeping": [], "hf_generic": [], "unknown": [], "ssm_unknown": []}
    for k in raw:
        if k in _CONSUMED_KEYS:
            out["consumed"].append(k)
        elif k in _BOOKKEEPING_KEYS:
            out["bookkeeping"].append(k)
        elif k in generic:
            out["hf_generic"].append(k)
        else:
            out["unknown"].append(k)
    for k in (raw.get("ssm_cfg") or {}):
        if k not in _SSM_FORWARD_KEYS and k not in _SSM_INIT_KEYS:
            out["ssm_unknown"].append(k)
    return out


def config_from_dict(raw: dict, strict: bool = False) -> "CaduceusConfig":
    """CaduceusConfig from the dict of a snapshot's config.json.  Unknown keys (top level or inside ssm_cfg) raise with strict=True
    and are logged as a warning otherwise; bookkeeping keys are dropped; the configuration must be one the engine implements."""
    import logging
    rep = audit_config_dict(raw)
    bad = rep["unknown"] + ["ssm_cfg." + k for k in rep["ssm_unknown"]]
    if bad:
        msg = ("config.json holds keys this implementation does not know and would ignore: %s (known: %s; ssm_cfg: %s)"
               % (bad, list(_CONSUMED_KEYS), list(_SSM_FORWARD_KEYS + _SSM_INIT_KEYS)))
        if strict:
            raise ValueError(msg)
        logging.warning(msg)
    kw = {k: v for k, v in raw.items() if k not in _BOOKKEEPING_KEYS}
    return CaduceusConfig(**kw)


# Published PlantCaduceus sizes (reference README.md:58-63); l20 dims confirmed by
# notebooks/examples.ipynb:66,74-79.  Used only to build synthetic checkpoints.
PLANTCADUCEUS_SIZES = {
    "l20": dict(d_model=384, n_layer=20),
    "l24": dict(d_model=512, n_layer=24),
    "l28": dict(d_model=768, n_layer=28),
    "l32": dict(d_model=1024, n_layer=32),
    # PlantCAD2 (reference docs/PlantCAD2-overview.md:19-21; 8 192-bp context, same block): dt_rank = ceil(d_model / 16) = 48 / 64 / 96
    "pc2-small": dict(d_model=768, n_layer=24),
    "pc2-medium": dict(d_model=1024, n_layer=48),
    "pc2-large": dict(d_model=1536, n_layer=48),
}
