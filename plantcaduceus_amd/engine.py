"""ctypes binding of libpcad.so (include/pcad.h) — PyTorch is plumbing only: device memory + streams.

The product path has NO CPU fallback: if the HIP library is missing, `load_library()` raises, and
`Engine.forward` refuses tensors that are not on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCAD_LIB") or os.path.join(_HERE, "libpcad.so")   # PCAD_LIB: developer A/B of two builds
CSRC = os.path.join(_HERE, "csrc")

PCAD_F32, PCAD_BF16 = 0, 1
_DT = {torch.float32: PCAD_F32, torch.bfloat16: PCAD_BF16}


class PcadConfig(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("n_layer", C.c_int32), ("d_state", C.c_int32), ("d_conv", C.c_int32),
        ("expand", C.c_int32), ("dt_rank", C.c_int32), ("vocab", C.c_int32), ("eps", C.c_float),
        ("dtype", C.c_int32), ("residual_in_fp32", C.c_int32), ("complement", C.c_int32 * 8),
    ]


class PcadKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double)]


class PcadTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


# name -> (restype, argtypes): every symbol include/pcad.h declares
SIGNATURES = {
    "pcad_version": (C.c_int, []),
    "pcad_last_error": (C.c_char_p, []),
    "pcad_build_hash": (C.c_char_p, []),
    "pcad_set_status_buffer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pcad_create": (C.c_int, [C.POINTER(PcadConfig), C.POINTER(C.c_void_p)]),
    "pcad_destroy": (None, [C.c_void_p]),
    "pcad_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pcad_weight_arena_bytes": (C.c_size_t, [C.c_void_p]),
    "pcad_bind_weights": (C.c_int, [C.c_void_p, C.POINTER(PcadTensor), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pcad_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "pcad_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pcad_forward_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "pcad_forward_all_hidden": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pcad_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pcad_profile_read": (C.c_int, [C.c_void_p, C.POINTER(PcadKernelStat), C.c_int]),
    "pcad_add_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                   C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "pcad_causal_conv1d_silu": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pcad_conv_xproj_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pcad_conv_xproj_bidir": (C.c_int, [C.c_void_p] * 14 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pcad_selective_scan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "pcad_selective_scan_dtproj": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pcad_gemm_nt": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pcad_gemm_nt_residual": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pcad_gemm_nt_split_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "pcad_gemm_nt_split": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "pcad_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                   C.c_void_p]),
    "pcad_final_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

_lib = None

STATUS_BAD_TOKEN, STATUS_BAD_POSITION = 1, 2       # include/pcad.h pcad_status_bits


def source_hash() -> str:
    """sha1 over csrc/*.hip, *.hpp (csrc/source_hash.py: the value the Makefile bakes into libpcad.so as pcad_build_hash)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_pcad_source_hash", os.path.join(CSRC, "source_hash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash(CSRC)


def build_library(force: bool = False) -> str:
    """Compile the HIP sources in-tree (`hipcc --offload-arch=gfx950`, cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j", "6"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libpcad.so failed:\n" + r.stdout[-4000:] + "\n" + r.stderr[-4000:])
    return LIB_PATH


def load_library():
    """Load libpcad.so; fails loudly if it is missing (no CPU fallback on the product path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C plantcaduceus_amd/csrc`). "
            "There is deliberately no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # a binary that was not built from the sources beside it would silently run (and be benchmarked as) other kernels
    if os.path.exists(os.path.join(CSRC, "source_hash.py")) and os.environ.get("PCAD_ALLOW_STALE") != "1":
        built, have = lib.pcad_build_hash().decode(), source_hash()
        if built != have:
            raise RuntimeError(
                f"{LIB_PATH} was built from other kernel sources (pcad_build_hash {built}, sources {have}): rebuild it with "
                "`make -C plantcaduceus_amd/csrc` / `__graft_entry__.build()` (PCAD_ALLOW_STALE=1 overrides, for A/B of old builds)")
    _lib = lib
    return lib


def _check(code: int, what: str):
    if code != 0:
        msg = load_library().pcad_last_error()
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a ROCm device (got {t.device}); the MI355X engine has no CPU path")


class Engine:
    """One handle = one model on one GPU.  Owns (as torch tensors) the weight arena and workspace."""

    def __init__(self, config, state_dict: Dict[str, torch.Tensor], dtype: torch.dtype, device: torch.device):
        if dtype not in _DT:
            raise ValueError(f"unsupported dtype {dtype}: the engine computes in bf16 or fp32")
        config.check_supported()
        self.lib = load_library()
        self.config = config
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the MI355X engine needs a ROCm device ('cuda:N'); there is no CPU path")
        cfg = PcadConfig(
            d_model=config.d_model, n_layer=config.n_layer, d_state=config.d_state, d_conv=config.d_conv,
            expand=config.expand, dt_rank=config.dt_rank, vocab=config.padded_vocab_size, eps=config.norm_epsilon,
            dtype=_DT[dtype], residual_in_fp32=int(bool(config.residual_in_fp32)),
            complement=(C.c_int32 * 8)(*config.complement_list()[:8]))
        self._h = C.c_void_p()
        _check(self.lib.pcad_create(C.byref(cfg), C.byref(self._h)), "pcad_create")
        self._ws: Optional[torch.Tensor] = None
        for key, val in (getattr(config, "engine_options", None) or {}).items():
            self.set_option(key, int(val))
        with torch.cuda.device(self.device):
            nbytes = self.lib.pcad_weight_arena_bytes(self._h)
            self._arena = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._bind(state_dict)
            # input-validation flags: a device word the kernels OR bits into + a pinned host mirror filled by an async copy
            with torch.inference_mode(False):      # normal tensors: they are updated in place from inside AND outside inference mode
                self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
                self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._status_event = torch.cuda.Event()
            self._status_event.record()
            _check(self.lib.pcad_set_status_buffer(self._h, self._status.data_ptr()), "pcad_set_status_buffer")

    def _aligned(self, t: torch.Tensor) -> int:
        return (t.data_ptr() + 255) // 256 * 256

    def _bind(self, sd: Dict[str, torch.Tensor]):
        keep = []
        arr = (PcadTensor * len(sd))()
        n = 0
        for name, t in sd.items():
            if not torch.is_tensor(t) or not t.is_floating_point():
                continue
            tt = t.detach()
            if tt.dtype not in _DT:
                tt = tt.float()
            tt = tt.to(self.device).contiguous()
            keep.append(tt)
            shape = list(tt.shape)[:4] + [1] * (4 - min(4, tt.dim()))
            if tt.dim() > 4:
                raise ValueError(name)
            arr[n] = PcadTensor(name.encode(), tt.data_ptr(), _DT[tt.dtype], min(4, tt.dim()), (C.c_int64 * 4)(*shape))
            n += 1
        base = self._aligned(self._arena)
        _check(self.lib.pcad_bind_weights(self._h, arr, n, base, self._arena.numel() - (base - self._arena.data_ptr()),
                                          _stream_ptr()), "pcad_bind_weights")
        torch.cuda.current_stream().synchronize()   # source tensors in `keep` may be freed after this
        del keep

    def _workspace(self, B: int, L: int):
        need = self.lib.pcad_workspace_bytes(self._h, B, L)
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = self._aligned(self._ws)
        return base, self._ws.numel() - (base - self._ws.data_ptr())

    def forward(self, input_ids: torch.Tensor, positions=None,
                want_hidden: bool = False, want_logits: bool = True, all_hidden: bool = False):
        """ids [B, L] (any int dtype, on this device) -> (logits fp32 [B,Q,8] | None, hidden [B,Q,2D] | None[, all]).
        positions: None (all L), a short list shared by every window, or an integer tensor [B] on this device
        (one position per window, Q = 1).

        Nothing here synchronises with the device.  Token ids outside the vocabulary and per-window positions outside the
        window - for which the reference raises an index error - are detected by the forward's last kernel and reported
        asynchronously: `check_status()` (which the host loops of this package call where they read results back) raises
        IndexError, and so does the next `forward` call once the flag of an earlier one has arrived."""
        _require_gpu(input_ids, "input_ids")
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [B, L], got {tuple(input_ids.shape)}")
        if input_ids.device != self.device:
            raise RuntimeError(f"input_ids on {input_ids.device}, engine on {self.device}")
        self._poll_status()
        ids = input_ids.to(torch.int32).contiguous()
        B, L = ids.shape
        D = self.config.d_model
        per_seq = None
        if torch.is_tensor(positions):
            if positions.dim() != 1 or positions.shape[0] != B or positions.device != self.device:
                raise ValueError("per-window positions must be an integer tensor [B] on the engine's device")
            per_seq = positions.to(torch.int32).contiguous()
            positions = None
        P = 0 if positions is None else len(positions)
        Q = 1 if per_seq is not None else (P if P else L)
        with torch.cuda.device(self.device):
            logits = torch.empty((B, Q, 8), dtype=torch.float32, device=self.device) if want_logits else None
            hidden = torch.empty((B, Q, 2 * D), dtype=self.dtype, device=self.device) if want_hidden else None
            if B == 0:
                return (logits, hidden, None) if all_hidden else (logits, hidden)
            ws, ws_bytes = self._workspace(B, L)
            lp = logits.data_ptr() if logits is not None else None
            hp = hidden.data_ptr() if hidden is not None else None
            try:
                if all_hidden:
                    if positions is not None or per_seq is not None:
                        raise ValueError("all_hidden requires positions=None")
                    allh = torch.empty((self.config.n_layer, B, L, 2 * D), dtype=self.dtype, device=self.device)
                    _check(self.lib.pcad_forward_all_hidden(self._h, ids.data_ptr(), B, L, allh.data_ptr(), hp, lp, ws,
                                                            ws_bytes, _stream_ptr()), "pcad_forward_all_hidden")
                    return logits, hidden, allh
                if per_seq is not None:
                    _check(self.lib.pcad_forward_at(self._h, ids.data_ptr(), B, L, per_seq.data_ptr(), hp, lp, ws, ws_bytes,
                                                    _stream_ptr()), "pcad_forward_at")
                    return logits, hidden
                pos_arr = (C.c_int32 * P)(*[int(p) for p in positions]) if P else None
                _check(self.lib.pcad_forward(self._h, ids.data_ptr(), B, L, pos_arr, P, hp, lp, ws, ws_bytes,
                                             _stream_ptr()), "pcad_forward")
            finally:
                # the status word follows the forward on the stream into pinned host memory; nobody waits for it here
                self._status_host.copy_(self._status, non_blocking=True)
                self._status_event.record()
        return logits, hidden

    # -- asynchronous input validation (include/pcad.h pcad_set_status_buffer) ----------------------------------------
    def _raise_status(self, bits: int):
        self._status.zero_()
        self._status_host.zero_()
        V = int(self.config.padded_vocab_size)
        what = []
        if bits & STATUS_BAD_TOKEN:
            what.append(f"input_ids contain token ids outside [0, {V}): check the tokenizer's vocabulary against the model")
        if bits & STATUS_BAD_POSITION:
            what.append("a per-window position is outside [0, L)")
        raise IndexError("; ".join(what) + " (detected on the device; results of that forward are invalid)")

    def _poll_status(self):
        """Non-blocking: raises if the status word of an EARLIER forward has already arrived and is set."""
        if self._status_event.query() and int(self._status_host[0]) != 0:
            self._raise_status(int(self._status_host[0]))

    def status_bits(self) -> int:
        """Blocking, non-raising: waits for the forwards enqueued so far and returns their accumulated status bits (0 = clean).
        Distributed host loops reduce this over the ranks first, so that every rank raises together (zero_shot.check_model_inputs)."""
        self._status_event.synchronize()
        return int(self._status_host[0])

    def check_status(self, bits: Optional[int] = None):
        """Blocking: waits for the forwards enqueued so far and raises IndexError if any of them saw an invalid token id or
        position (the reference's nn.Embedding / indexing errors).  Host loops call this where they read results back.
        bits: status bits already collected (e.g. OR-ed over the ranks of a process group) instead of this engine's own."""
        if bits is None:
            bits = self.status_bits()
        if bits:
            self._raise_status(bits)

    def set_option(self, key: str, value: int):
        """`pcad_set_option` (include/pcad.h): "chunk_seqs" (windows per pass through the stack), "gate_each" (reference-order
        SiLU gate), "norm_fold" (add + RMSNorm folded into out_proj's epilogue / in_proj), "reference_order" (0 / 1 / 2: one switch for
        the reference's rounding points), "f32_gemm_split" (fp32 model: split-bf16 in_proj / out_proj), "scan_segments" (segmented
        scan of long strands), "last_layer_shortcut", "poison_workspace" (debug).  "norm_fold" 1 on an fp32 model and
        "f32_gemm_split" need weight copies packed at bind time: pass them as `config.engine_options` (applied before binding)."""
        _check(self.lib.pcad_set_option(self._h, key.encode(), int(value)), "pcad_set_option")
        self._ws = None

    def release_workspace(self):
        """Free the workspace slab (it is re-allocated by the next forward): for a process that keeps several models resident
        and runs them in turn.  `set_option("workspace_limit_mb", N)` bounds what a forward allocates in the first place."""
        self._ws = None

    def profile(self, on):
        """False/0: off; True/1: HIP events around every launch; N > 1: around every N-th launch of each kernel class."""
        _check(self.lib.pcad_profile_enable(self._h, int(on)), "pcad_profile_enable")

    def profile_read(self):
        """-> {kernel class: (launches, total_ms)} since the last read (waits for the events)."""
        arr = (PcadKernelStat * 16)()
        n = self.lib.pcad_profile_read(self._h, arr, 16)
        if n < 0:
            _check(n, "pcad_profile_read")
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(n)}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.pcad_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
