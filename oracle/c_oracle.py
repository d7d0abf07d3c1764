"""ctypes wrapper of oracle/c/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/c/pcad_oracle.c).

Used by tests/ (cross-check against the torch oracle and against the HIP path at sizes the torch
oracle is too slow for), by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _host_tag() -> str:
    """The library is compiled -march=native, so key its file name by this host's CPU feature flags."""
    import hashlib
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


LIB = os.path.join(_HERE, "c", "liboracle-%s.so" % _host_tag())
FP = C.POINTER(C.c_float)


class OracleLayer(C.Structure):
    _fields_ = [("norm_w", FP), ("in_proj", FP), ("out_proj", FP), ("conv_w", FP * 2), ("conv_b", FP * 2),
                ("x_proj", FP * 2), ("dt_w", FP * 2), ("dt_b", FP * 2), ("A_log", FP * 2), ("Dskip", FP * 2)]


class OracleModel(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_layer", C.c_int32), ("d_inner", C.c_int32), ("dt_rank", C.c_int32),
                ("eps", C.c_float), ("emulate_bf16", C.c_int32), ("ref_order", C.c_int32), ("complement", C.c_int32 * 8), ("emb", FP), ("norm_f", FP),
                ("layers", C.POINTER(OracleLayer))]


def build():
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "c"), "TARGET=" + os.path.basename(LIB)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)


def _lib():
    build()   # make: no-op when this host's copy is up to date
    lib = C.CDLL(LIB)
    lib.oracle_forward.restype = C.c_int
    lib.oracle_forward.argtypes = [C.POINTER(OracleModel), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_num_threads.restype = C.c_int
    lib.omp_set_num_threads.restype = None          # libgomp's, reached through the library's own dependency (dlsym on its handle)
    lib.omp_set_num_threads.argtypes = [C.c_int]
    lib.oracle_set_gemm.restype = None
    lib.oracle_set_gemm.argtypes = [C.c_void_p]
    return lib


def usable_cpus(read=None, affinity=None):
    """CPUs this process may actually use: its affinity mask, capped by the container's CFS quota (cgroup v2 cpu.max / v1
    cpu.cfs_quota_us).  The GPU boxes of this pool show 256 CPUs under a quota of 16; a larger team is throttled as a group and its
    spinning waiters burn the quota (profiles/r06_host_probe.txt: 0.8 -> 2.8 windows/s at l32).  None when there is no quota."""
    def rd(path):
        if read is not None:           # tests: a dict of file contents
            return read.get(path, "").strip()
        try:
            return open(path).read().strip()
        except OSError:
            return ""
    quota = None
    try:
        v2 = rd("/sys/fs/cgroup/cpu.max").split()
        if len(v2) == 2 and v2[0] != "max":
            quota = float(v2[0]) / float(v2[1])
        elif not v2:
            q, p = rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
            if q and p and float(q) > 0 and float(p) > 0:
                quota = float(q) / float(p)
    except ValueError:
        quota = None
    if quota is None:
        return None
    try:
        aff = affinity or len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    n = max(1, min(aff, int(quota + 0.999)))
    return n if n < aff else None


GEMM_FN = C.CFUNCTYPE(None, FP, C.c_int, FP, C.c_int, FP, C.c_int, C.c_int, C.c_int)


@GEMM_FN
def _blas_gemm(A, lda, W, K, Cp, ldc, M, N):
    """C[M, N] = A[M, :K] . W[N, K]^T through numpy (the host's BLAS sgemm, all cores)."""
    a = np.ctypeslib.as_array(A, shape=(M, lda))[:, :K]
    w = np.ctypeslib.as_array(W, shape=(N, K))
    c = np.ctypeslib.as_array(Cp, shape=(M, ldc))[:, :N]
    np.matmul(a, w.T, out=c)


@GEMM_FN
def _torch_gemm(A, lda, W, K, Cp, ldc, M, N):
    """the same product through torch.mm (the ATen CPU GEMM: MKL / oneDNN / its own OpenBLAS, whichever this torch build carries)."""
    import torch
    a = torch.from_numpy(np.ctypeslib.as_array(A, shape=(M, lda)))[:, :K]
    w = torch.from_numpy(np.ctypeslib.as_array(W, shape=(N, K)))
    c = torch.from_numpy(np.ctypeslib.as_array(Cp, shape=(M, ldc)))[:, :N]
    if c.is_contiguous():
        torch.mm(a, w.t(), out=c)
    else:
        c.copy_(torch.mm(a, w.t()))


class COracle:
    """fp32 C/OpenMP forward from a reference-named state dict (values rounded through `dtype` first,
    emulating from_pretrained(torch_dtype=...); arithmetic is always fp32).  emulate_bf16=True additionally rounds
    to bf16 at the tensor boundaries where the reference's bf16 model stores a bf16 tensor (u, xz, conv out, x_dbl,
    delta, each direction's gated scan output and their sum, out_proj out, hidden, logits) — the engine's order,
    i.e. caduceus_oracle.forward_strands(rnd=round_bf16, tie_fold=True).  ref_order=True keeps the reference's order of
    the tied out_proj instead (each direction projected and rounded, then summed: forward_strands(tie_fold=False)); in
    both orders each direction's scan output is gated by SiLU(z) and rounded separately, as the reference's two
    selective_scan_fn calls do."""

    def __init__(self, state_dict, config, dtype=None, emulate_bf16=False, ref_order=False, blas=False, threads=None):
        import torch
        self.lib = _lib()
        # threads: team size of the OpenMP loops AND of the library GEMM.  None: each runtime's own default, unless the container has a
        # CPU quota below the CPUs it shows - then the quota (usable_cpus), since a team above it is throttled and spinning waiters burn it.
        # Process-wide (omp_set_num_threads of the calling thread): later COracle objects of the process inherit it
        self.nthreads = int(threads) if threads else usable_cpus()
        if self.nthreads:
            self.lib.omp_set_num_threads(self.nthreads)
        # the four projections through a host library GEMM instead of the plain-C one: True / "numpy" = numpy's sgemm, "torch" = torch.mm
        self.blas = {False: None, None: None, True: _blas_gemm, "numpy": _blas_gemm, "torch": _torch_gemm}[blas]
        self.config = config
        self._keep = []

        def arr(name):
            t = state_dict[name]
            if dtype is not None:
                t = t.to(dtype)
            a = np.ascontiguousarray(t.float().cpu().numpy(), dtype=np.float32)
            self._keep.append(a)
            return a.ctypes.data_as(FP)

        pre = "caduceus.backbone."
        self.layers = (OracleLayer * config.n_layer)()
        for i in range(config.n_layer):
            lp = f"{pre}layers.{i}."
            ly = self.layers[i]
            ly.norm_w = arr(lp + "norm.weight")
            ly.in_proj = arr(lp + "mixer.submodule.mamba_fwd.in_proj.weight")
            ly.out_proj = arr(lp + "mixer.submodule.mamba_fwd.out_proj.weight")
            for d, nm in enumerate(("fwd", "rev")):
                mp = f"{lp}mixer.submodule.mamba_{nm}."
                ly.conv_w[d] = arr(mp + "conv1d.weight")
                ly.conv_b[d] = arr(mp + "conv1d.bias")
                ly.x_proj[d] = arr(mp + "x_proj.weight")
                ly.dt_w[d] = arr(mp + "dt_proj.weight")
                ly.dt_b[d] = arr(mp + "dt_proj.bias")
                ly.A_log[d] = arr(mp + "A_log")
                ly.Dskip[d] = arr(mp + "D")
        self.model = OracleModel(
            d_model=config.d_model, n_layer=config.n_layer, d_inner=config.d_inner, dt_rank=config.dt_rank,
            eps=config.norm_epsilon, emulate_bf16=int(bool(emulate_bf16)), ref_order=int(bool(ref_order)), complement=(C.c_int32 * 8)(*config.complement_list()[:8]),
            emb=arr(pre + "embeddings.word_embeddings.embedding.weight"), norm_f=arr(pre + "norm_f.weight"),
            layers=self.layers)

    @property
    def threads(self) -> int:
        return int(self.lib.oracle_num_threads())

    def forward(self, ids, want_logits=True, want_hidden=False):
        ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int32)
        B, L = ids.shape
        logits = np.empty((B, L, 8), dtype=np.float32) if want_logits else None
        hidden = np.empty((B, L, 2 * self.config.d_model), dtype=np.float32) if want_hidden else None
        self.lib.oracle_set_gemm(C.cast(self.blas, C.c_void_p) if self.blas else None)
        import contextlib
        limit = contextlib.nullcontext()
        if self.nthreads and self.blas is _blas_gemm:
            import threadpoolctl
            limit = threadpoolctl.threadpool_limits(limits=self.nthreads, user_api="blas")
        elif self.nthreads and self.blas is _torch_gemm:
            import torch
            torch.set_num_threads(self.nthreads)
        try:
            with limit:
                rc = self.lib.oracle_forward(C.byref(self.model), ids.ctypes.data, B, L,
                                             logits.ctypes.data if want_logits else None,
                                             hidden.ctypes.data if want_hidden else None)
        finally:
            self.lib.oracle_set_gemm(None)
        if rc != 0:
            raise MemoryError("oracle_forward failed")
        return logits, hidden


class COracleForMaskedLM:
    """HF-surface stand-in over the C port (`model(input_ids=..., output_hidden_states=...)` -> `.logits`,
    `.hidden_states[-1]`), so the reference-shaped CLIs can be driven on CPU at real model sizes (BASELINE config 1)."""

    def __init__(self, state_dict, config, **kw):
        self.oracle = COracle(state_dict, config, **kw)
        self.config = config

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, input_ids=None, output_hidden_states=False, **kw):
        import torch
        from oracle.caduceus_oracle import _Out
        ids = input_ids.cpu().numpy() if hasattr(input_ids, "cpu") else np.asarray(input_ids)
        lg, hid = self.oracle.forward(ids, want_logits=True, want_hidden=bool(output_hidden_states))
        return _Out(torch.from_numpy(lg), (torch.from_numpy(hid),) if output_hidden_states else None)
