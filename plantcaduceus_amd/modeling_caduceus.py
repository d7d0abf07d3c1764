"""HF `AutoModel` / `AutoModelForMaskedLM` surface over the MI355X engine.

Drop-in for the two reference call sites of the hot path:
  * `model = AutoModelForMaskedLM.from_pretrained(dir, trust_remote_code=True, torch_dtype=dtype)`,
    `model.to(device)`, `model(input_ids=ids).logits`                    (reference src/zero_shot_score.py:91-97,115-118)
  * `model(input_ids=ids, output_hidden_states=True).hidden_states[-1]`  (reference src/train_XGBoost.py:104-105)
  * notebook usage with `device_map=device` and `AutoModel` -> `.last_hidden_state`
    (reference notebooks/examples.ipynb:109-112,169-170).

The module tree reproduces the reference's parameter names (notebooks/examples.ipynb:61-100) so
`state_dict()` keys equal those of a real `kuleshov-group/PlantCaduceus_l*` snapshot; the modules
are parameter holders only — the arithmetic runs in libpcad.so (HIP, gfx950) through `engine.Engine`.
There is no CPU execution path: calling the model with CPU tensors raises.
"""
from __future__ import annotations

import json
import os
from typing import Optional, Sequence

import torch
from torch import nn
from transformers import PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithNoAttention, MaskedLMOutput

from .checkpoint import load_state_dict, resolve_snapshot
from .configuration_caduceus import CaduceusConfig, config_from_dict
from .engine import Engine


# ---- parameter holders with the reference's names --------------------------------------------------
class _Weight(nn.Module):
    def __init__(self, *shape, bias_shape=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape), requires_grad=False)
        if bias_shape is not None:
            self.bias = nn.Parameter(torch.empty(*bias_shape), requires_grad=False)


class _Mamba(nn.Module):
    def __init__(self, cfg: CaduceusConfig):
        super().__init__()
        D, E, N, R, W = cfg.d_model, cfg.d_inner, cfg.d_state, cfg.dt_rank, cfg.d_conv
        self.in_proj = _Weight(2 * E, D)
        self.conv1d = _Weight(E, 1, W, bias_shape=(E,))
        self.x_proj = _Weight(R + 2 * N, E)
        self.dt_proj = _Weight(E, R, bias_shape=(E,))
        self.A_log = nn.Parameter(torch.empty(E, N), requires_grad=False)
        self.D = nn.Parameter(torch.empty(E), requires_grad=False)
        self.out_proj = _Weight(cfg.d_model, E)


class _BiMamba(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.mamba_fwd = _Mamba(cfg)
        self.mamba_rev = _Mamba(cfg)
        # bidirectional_weight_tie (BiMambaWrapper): in_proj / out_proj shared
        self.mamba_rev.in_proj.weight = self.mamba_fwd.in_proj.weight
        self.mamba_rev.out_proj.weight = self.mamba_fwd.out_proj.weight


class _RCPSWrapper(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.submodule = _BiMamba(cfg)


class _Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.mixer = _RCPSWrapper(cfg)
        self.norm = _Weight(cfg.d_model)


class _RCPSEmbedding(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embedding = _Weight(cfg.padded_vocab_size, cfg.d_model)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = _RCPSEmbedding(cfg)


class _MixerModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.layers = nn.ModuleList([_Block(cfg) for _ in range(cfg.n_layer)])
        self.norm_f = _Weight(cfg.d_model)


class _LMHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.lm_head = _Weight(cfg.padded_vocab_size, cfg.d_model)


class LastHiddenOnly:
    """`hidden_states` when only the last level was materialised (the default: the reference's callers read `[-1]` only and
    all n_layer+1 levels at B=1024 / l32 are 71 GB).  A plain sequence-like object (NOT a tuple subclass: helpers that rebuild
    containers with `type(obj)(generator)` — accelerate's `honor_type`, copy, pickle — must not mistake it for a tuple of
    n_layer + 1 tensors): `len()` is n_layer + 1, `[-1]` / `[n_layer]` is the final hidden state, and any level that was not kept
    raises, loudly.  Not supported: iteration, slicing, `torch.stack(hidden_states)`, accelerate device_map hooks that walk
    outputs — set `config.materialize_all_hidden_states = True` to get the real tuple for those."""

    __slots__ = ("last", "_n")

    def __init__(self, last, n_levels: int):
        self.last = last
        self._n = int(n_levels)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            raise IndexError("only hidden_states[-1] is materialised; set config.materialize_all_hidden_states = True for slices")
        j = i + self._n if i < 0 else i
        if j == self._n - 1:
            return self.last
        if 0 <= j < self._n:
            raise IndexError(f"hidden_states[{i}] was not materialised (only [-1] is, by default); set "
                             "config.materialize_all_hidden_states = True to get all n_layer + 1 levels")
        raise IndexError("tuple index out of range")

    def __iter__(self):
        raise TypeError("only hidden_states[-1] is materialised; set config.materialize_all_hidden_states = True to iterate")

    def __reduce__(self):
        return (LastHiddenOnly, (self.last, self._n))

    def to(self, *args, **kwargs):
        return LastHiddenOnly(self.last.to(*args, **kwargs), self._n)

    def __repr__(self):
        return f"LastHiddenOnly(levels={self._n}, last={tuple(self.last.shape)})"


# ---- HF models --------------------------------------------------------------------------------------
class CaduceusPreTrainedModel(PreTrainedModel):
    config_class = CaduceusConfig
    base_model_prefix = "caduceus"
    supports_gradient_checkpointing = False
    supports_positions = True     # forward(..., positions=[p, ...]) evaluates the head at those rows only
    _no_split_modules = ["_Block"]

    def _init_weights(self, module):   # weights always come from a checkpoint
        pass

    # -- loading: own reader (config.json + safetensors/bin with the reference key names) ------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, torch_dtype=None, dtype=None,
                        device_map=None, trust_remote_code=None, **kwargs):
        hub_kw = {k: kwargs[k] for k in ("revision", "cache_dir", "local_files_only", "token") if k in kwargs}
        path = resolve_snapshot(pretrained_model_name_or_path, **hub_kw)     # directory, or a hub id via the HF cache
        if config is None:
            with open(os.path.join(path, "config.json")) as f:
                raw = json.load(f)
            config = config_from_dict(raw)          # warns about keys it would otherwise ignore silently (strict audit: tools/real_weights.sh)
        want = dtype if dtype is not None else torch_dtype
        if isinstance(want, str):
            want = getattr(torch, want) if want != "auto" else None
        if want is None:
            want = torch.float32
        model = cls(config)
        sd = load_state_dict(path)
        if cls.base_model_prefix and not hasattr(model, cls.base_model_prefix):
            # backbone-only class (AutoModel): snapshot keys carry the "caduceus." prefix
            pre = cls.base_model_prefix + "."
            sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        own = model.state_dict()
        missing = [k for k in own if k not in sd and not k.startswith("lm_head.")]
        if missing:
            raise KeyError(f"checkpoint {path} lacks {len(missing)} tensors, e.g. {missing[:3]}")
        for k, p in own.items():
            src = sd.get(k)
            if src is None:
                continue
            if tuple(src.shape) != tuple(p.shape):
                raise ValueError(f"{k}: checkpoint shape {tuple(src.shape)} != model {tuple(p.shape)}")
            p.copy_(src)
        model.tie_weights()
        model.to(want)
        model.eval()
        if device_map is not None:
            dev = device_map if not isinstance(device_map, dict) else next(iter(device_map.values()))
            if dev not in ("auto", None):
                model.to(dev)
        return model

    def preferred_batch_size(self, seqlen: int) -> int:
        """Windows per forward call the host loops batch up to when the user gave no `-batchSize`: two chunks of 2^31 bytes per
        [rows, d_inner] tensor (1024 windows of 512 bp at l32 bf16 = the benchmark's batch, two 512-window chunks of the
        layer-stack walk sharing one 15.3 GB workspace; see csrc/api.hip — the engine's own cap per chunk is (2^32 - 2 MiB) bytes,
        and it splits a batch evenly into the fewest chunks), halved until the workspace the engine would really allocate for it
        (`pcad_workspace_bytes`, i.e. one chunk) fits in a third of the device's free memory."""
        p = self._backbone_owner().caduceus_param()
        L = max(1, int(seqlen))
        rows = (1 << 31) // (self.config.d_inner * p.element_size())
        want = max(1, 2 * (rows // (2 * L)))
        if p.device.type == "cuda":
            try:
                free, _ = torch.cuda.mem_get_info(p.device)
                eng = self._engine()
                while want > 1 and eng.lib.pcad_workspace_bytes(eng._h, want, L) > free // 3:
                    want //= 2
            except Exception:
                pass
        return want

    def check_status(self, bits=None):
        """Deferred input validation of the engine (token ids / positions outside their range raise IndexError here).
        bits: status bits to raise for instead of this engine's own (the OR over the ranks of a process group)."""
        eng = getattr(self._backbone_owner(), "_pcad_engine", None)
        if eng is not None:
            eng.check_status(bits)

    def status_bits(self) -> int:
        """The engine's accumulated input-validation bits (blocking, non-raising; 0 when no forward ran yet)."""
        eng = getattr(self._backbone_owner(), "_pcad_engine", None)
        return eng.status_bits() if eng is not None else 0

    # -- engine plumbing ----------------------------------------------------------------------------
    def _backbone_owner(self):
        raise NotImplementedError

    def _engine(self) -> Engine:
        owner = self._backbone_owner()
        p = owner.caduceus_param()
        key = (p.device, p.dtype, p.data_ptr(), p._version)
        eng = getattr(owner, "_pcad_engine", None)
        if eng is None or getattr(owner, "_pcad_key", None) != key:
            if p.device.type != "cuda":
                raise RuntimeError(
                    f"model parameters are on {p.device}: the MI355X engine needs a ROCm device "
                    "(model.to('cuda:0')); there is deliberately no CPU fallback on the product path")
            sd = {k: v for k, v in owner.full_state_dict().items()}
            eng = Engine(self.config, sd, p.dtype, p.device)
            owner._pcad_engine = eng
            owner._pcad_key = key
        return eng


class Caduceus(CaduceusPreTrainedModel):
    """`AutoModel` class: backbone only; `.last_hidden_state` is [B, L, 2*d_model]."""

    def __init__(self, config: CaduceusConfig, **kwargs):
        super().__init__(config)
        self.backbone = _MixerModel(config)
        self._pcad_engine = None

    def _backbone_owner(self):
        return self

    def caduceus_param(self):
        return self.backbone.embeddings.word_embeddings.embedding.weight

    def full_state_dict(self):
        return {"caduceus." + k: v for k, v in self.state_dict().items()}

    def tie_weights(self, *a, **k):
        pass

    def forward(self, input_ids=None, inputs_embeds=None, output_hidden_states=None, return_dict=None,
                positions: Optional[Sequence[int]] = None, **kwargs):
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds is not supported by the MI355X engine")
        eng = self._engine()
        if output_hidden_states and getattr(self.config, "materialize_all_hidden_states", False):
            _, last, allh = eng.forward(input_ids, want_hidden=True, want_logits=False, all_hidden=True)
            hs = tuple(allh[i] for i in range(allh.shape[0])) + (last,)
        else:
            _, last = eng.forward(input_ids, positions=positions, want_hidden=True, want_logits=False)
            hs = LastHiddenOnly(last, self.config.n_layer + 1) if output_hidden_states else None
        if return_dict is False:
            return (last, hs) if hs is not None else (last,)
        return BaseModelOutputWithNoAttention(last_hidden_state=last, hidden_states=hs)


class CaduceusForMaskedLM(CaduceusPreTrainedModel):
    """`AutoModelForMaskedLM` class: `.logits` fp32 [B, L, 8]; `.hidden_states[-1]` [B, L, 2*d_model].

    By default only `hidden_states[-1]` is materialised when `output_hidden_states=True` (the only entry
    the reference's callers read; all 33 levels at B=1024/l32 would be 71 GB): `hidden_states` is then a
    `LastHiddenOnly` — `len()` n_layer + 1 and `[-1]` as in HF, any other level raises.  Set
    `config.materialize_all_hidden_states = True` for the full n_layer+1 tuple.
    Extra (non-HF) keyword `positions=[p, ...]` evaluates the head only at those positions
    (logits [B, P, 8]) — the engine's fast path for zero-shot scoring.
    """

    def __init__(self, config: CaduceusConfig, **kwargs):
        super().__init__(config)
        self.caduceus = Caduceus(config)
        self.lm_head = _LMHead(config)
        self.tie_weights()

    def tie_weights(self, *a, **k):
        self.lm_head.lm_head.weight = self.caduceus.backbone.embeddings.word_embeddings.embedding.weight

    def _backbone_owner(self):
        return self.caduceus

    def get_input_embeddings(self):
        return self.caduceus.backbone.embeddings.word_embeddings.embedding

    def forward(self, input_ids=None, inputs_embeds=None, labels=None, output_hidden_states=None,
                return_dict=None, positions: Optional[Sequence[int]] = None, **kwargs):
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds is not supported by the MI355X engine")
        if labels is not None:
            raise NotImplementedError("this is an inference engine: loss/labels are not supported")
        eng = self._engine()
        hs = None
        if output_hidden_states and getattr(self.config, "materialize_all_hidden_states", False):
            logits, last, allh = eng.forward(input_ids, want_hidden=True, want_logits=True, all_hidden=True)
            hs = tuple(allh[i] for i in range(allh.shape[0])) + (last,)
        else:
            logits, last = eng.forward(input_ids, positions=positions, want_hidden=bool(output_hidden_states),
                                       want_logits=True)
            if output_hidden_states:
                hs = LastHiddenOnly(last, self.config.n_layer + 1)
        if return_dict is False:
            return (logits, hs) if hs is not None else (logits,)
        return MaskedLMOutput(loss=None, logits=logits, hidden_states=hs)


def register_auto_classes():
    """Make `AutoConfig/AutoModel/AutoModelForMaskedLM/AutoTokenizer.from_pretrained(dir)` resolve
    `model_type == "caduceus"` to this package (instead of the HF-hub remote code)."""
    from transformers import AutoConfig, AutoModel, AutoModelForMaskedLM, AutoTokenizer
    from .tokenization_caduceus import CaduceusTokenizer
    try:
        AutoConfig.register("caduceus", CaduceusConfig)
    except ValueError:
        pass
    for auto, klass in ((AutoModel, Caduceus), (AutoModelForMaskedLM, CaduceusForMaskedLM)):
        try:
            auto.register(CaduceusConfig, klass)
        except ValueError:
            pass
    try:
        AutoTokenizer.register(CaduceusConfig, slow_tokenizer_class=CaduceusTokenizer)
    except (ValueError, TypeError):
        pass
