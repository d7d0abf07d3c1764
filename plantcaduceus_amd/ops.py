"""Operator-level mirrors of the third-party functions the reference's hot path calls, running on the
MI355X HIP kernels through the C ABI.  Same names, argument meaning and layouts as the originals
 so the parity tests read like the upstream operators' own tests:

  rms_norm_fn(x, weight, bias, residual, eps, prenorm, residual_in_fp32)   mamba_ssm.ops.triton.layer_norm
  causal_conv1d_fn(x, weight, bias, activation)                            causal_conv1d 1.4.0
  selective_scan_fn(u, delta, A, B, C, D, z, delta_bias, delta_softplus)   mamba_ssm 2.2.2
  mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, ...)
                                                                           mamba_ssm 2.2.2 (the call Mamba.forward's fast path makes)
  linear(x, weight)                                                        F.linear (no bias)

Inputs use the reference's channels-first (B, E, L) layout where the originals do; they are transposed
to the engine's token-major layout here (test plumbing — the engine itself never transposes).
All tensors must be ROCm tensors; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import _DT, _check, _require_gpu, _stream_ptr, load_library


def _dt(t: torch.Tensor) -> int:
    if t.dtype not in _DT:
        raise ValueError(f"unsupported dtype {t.dtype}")
    return _DT[t.dtype]


def rms_norm_fn(x, weight, bias=None, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False):
    _require_gpu(x, "x")
    if bias is not None:
        raise NotImplementedError("RMSNorm bias is not used by Caduceus")
    lib = load_library()
    D = x.shape[-1]
    xf = x.contiguous().view(-1, D)
    rows = xf.shape[0]
    rdt = torch.float32 if (residual_in_fp32 or x.dtype == torch.float32) else x.dtype
    res_in = None
    if residual is not None:
        res_in = residual.to(rdt).contiguous().view(-1, D)
    y = torch.empty_like(xf)
    res_out = torch.empty((rows, D), dtype=rdt, device=x.device)
    w = weight.float().contiguous()
    with torch.cuda.device(x.device):
        _check(lib.pcad_add_rmsnorm(xf.data_ptr(), res_in.data_ptr() if res_in is not None else None, w.data_ptr(),
                                    y.data_ptr(), res_out.data_ptr(), rows, D, float(eps), _dt(xf), _DT[rdt],
                                    _stream_ptr()), "pcad_add_rmsnorm")
    y = y.view(x.shape)
    return (y, res_out.view(x.shape)) if prenorm else y


def causal_conv1d_bidir(x_tm, w_fwd, b_fwd, w_rev, b_rev):
    """Token-major both-direction form used by the engine: x_tm [S, L, E] -> (y_fwd, y_rev) [S, L, E]."""
    _require_gpu(x_tm, "x")
    lib = load_library()
    S, L, E = x_tm.shape
    x_tm = x_tm.contiguous()
    yf = torch.empty_like(x_tm)
    yr = torch.empty_like(x_tm)
    args = [t.float().contiguous() for t in (w_fwd.reshape(E, -1), b_fwd, w_rev.reshape(E, -1), b_rev)]
    with torch.cuda.device(x_tm.device):
        _check(lib.pcad_causal_conv1d_silu(x_tm.data_ptr(), E, args[0].data_ptr(), args[1].data_ptr(),
                                           args[2].data_ptr(), args[3].data_ptr(), yf.data_ptr(), yr.data_ptr(),
                                           S, L, E, _dt(x_tm), _stream_ptr()), "pcad_causal_conv1d_silu")
    return yf, yr


def to_blocked(x_rows: torch.Tensor) -> torch.Tensor:
    """[rows, E] -> the engine's blocked layout [rows8/8, E*esz/128, 8, 128/esz] (include/pcad.h), rows zero-padded to 8."""
    rows, E = x_rows.shape
    per = 128 // x_rows.element_size()
    if E % per:
        raise ValueError("E * elem must be a multiple of 128 bytes")
    rows8 = (rows + 7) // 8 * 8
    buf = torch.zeros((rows8, E), dtype=x_rows.dtype, device=x_rows.device)
    buf[:rows] = x_rows
    return buf.view(rows8 // 8, 8, E // per, per).permute(0, 2, 1, 3).contiguous()


def from_blocked(xb: torch.Tensor, rows: int) -> torch.Tensor:
    nb, pieces, _, per = xb.shape
    return xb.permute(0, 2, 1, 3).reshape(nb * 8, pieces * per)[:rows].contiguous()


def _conv_xproj_raw(x_tm, w_fwd, b_fwd, w_rev, b_rev, x_proj_fwd, x_proj_rev):
    """pcad_conv_xproj_bidir on a token-major x [S, L, E]: -> (xc [2] blocked, dtl [2] [rows, Rp], bc [2] fp32 [rows, 32], R, Rp)."""
    _require_gpu(x_tm, "x")
    lib = load_library()
    S, L, E = x_tm.shape
    dt, dev = x_tm.dtype, x_tm.device
    R = x_proj_fwd.shape[0] - 32
    if not 0 < R <= 96:
        raise ValueError("dt_rank must be in [1, 96] for the fused kernel")
    Rp = 64 if R <= 64 else 96
    rows = S * L
    xb = to_blocked(x_tm.reshape(rows, E))

    def pack_wx(w):
        p = torch.zeros((Rp + 32, E), dtype=dt, device=dev)
        p[:R] = w[:R].to(dt)
        p[Rp:] = w[R:].to(dt)
        return p
    wx = [pack_wx(x_proj_fwd), pack_wx(x_proj_rev)]
    taps = [t.float().contiguous() for t in (w_fwd.reshape(E, -1), b_fwd, w_rev.reshape(E, -1), b_rev)]
    scratch = torch.empty(lib.pcad_conv_xproj_scratch_bytes(E, _DT[dt]) + 256, dtype=torch.uint8, device=dev)
    xc = [torch.zeros_like(xb) for _ in range(2)]
    dtl = [torch.empty((rows, Rp), dtype=dt, device=dev) for _ in range(2)]
    bc = [torch.empty((rows, 32), dtype=torch.float32, device=dev) for _ in range(2)]
    with torch.cuda.device(dev):
        _check(lib.pcad_conv_xproj_bidir(xb.data_ptr(), taps[0].data_ptr(), taps[1].data_ptr(), taps[2].data_ptr(),
                                         taps[3].data_ptr(), wx[0].data_ptr(), wx[1].data_ptr(),
                                         (scratch.data_ptr() + 255) // 256 * 256,
                                         xc[0].data_ptr(), dtl[0].data_ptr(), bc[0].data_ptr(),
                                         xc[1].data_ptr(), dtl[1].data_ptr(), bc[1].data_ptr(),
                                         S, L, E, Rp, _DT[dt], _stream_ptr()), "pcad_conv_xproj_bidir")
    return xc, dtl, bc, R, Rp


def conv_xproj_bidir(x_tm, w_fwd, b_fwd, w_rev, b_rev, x_proj_fwd, x_proj_rev):
    """The engine's fused head of mamba_inner_fn, both directions from one read of x:
        xc_d    = causal_conv1d_fn(x, w_d, b_d, "silu")        (d = rev: anti-causal on the same rows)
        x_dbl_d = F.linear(xc_d, x_proj_d)                     (stored in the model dtype)
    x_tm [S, L, E]; w_* [E, 4]; b_* [E]; x_proj_* [R + 32, E] with R <= 96.
    Returns (xc_fwd, xc_rev [S, L, E], x_dbl_fwd, x_dbl_rev [S, L, R + 32]) in x's dtype."""
    S, L, E = x_tm.shape
    dt = x_tm.dtype
    rows = S * L
    xc, dtl, bc, R, Rp = _conv_xproj_raw(x_tm, w_fwd, b_fwd, w_rev, b_rev, x_proj_fwd, x_proj_rev)
    outs = [from_blocked(c, rows).view(S, L, E) for c in xc]
    dbl = [torch.cat([dtl[d][:, :R], bc[d].to(dt)], dim=1).view(S, L, R + 32) for d in range(2)]
    return outs[0], outs[1], dbl[0], dbl[1]


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x (B, E, L), weight (E, W=4), bias (E), activation must be "silu"."""
    if activation not in ("silu", "swish"):
        raise NotImplementedError("only activation='silu' is on the Caduceus path")
    if weight.shape[-1] != 4:
        raise NotImplementedError("only conv width 4")
    if bias is None:
        bias = torch.zeros(weight.shape[0], device=x.device)
    yf, _ = causal_conv1d_bidir(x.transpose(1, 2), weight, bias, weight, bias)
    return yf.transpose(1, 2)


def _scan_common(u, B, C, A, D, z, delta_bias):
    Bsz, E, L = u.shape
    dt = u.dtype
    u_tm = u.transpose(1, 2).contiguous()
    z_tm = z.to(dt).transpose(1, 2).contiguous() if z is not None else None
    # B_t | C_t rows in fp32 holding the values at the model dtype's precision (what the x_proj epilogue writes)
    bc = torch.cat([B.to(dt).transpose(1, 2), C.to(dt).transpose(1, 2)], dim=-1).float().contiguous()   # [B, L, 32]
    A32 = A.float().contiguous()
    Dv = (D.float() if D is not None else torch.zeros(E, device=u.device)).contiguous()
    db = (delta_bias.float() if delta_bias is not None else torch.zeros(E, device=u.device)).contiguous()
    return u_tm, z_tm, bc, A32, Dv, db


def _acc_mode(accumulate_into, gate_sum) -> int:
    if accumulate_into is None:
        if gate_sum:
            raise ValueError("gate_sum needs accumulate_into (the other direction's ungated output)")
        return 0
    return 2 if gate_sum else 1


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False, reverse=False, accumulate_into=None, gate_sum=False):
    """u, delta, z: (B, E, L); A: (E, 16); B, C: (B, 16, L); D, delta_bias: (E).  Returns (B, E, L).
    accumulate_into: add this call's gated output to an earlier one (bi-directional "add"); with gate_sum=True the
    earlier output is taken as UNGATED (its call had z=None) and SiLU(z) is applied once to the sum (the engine's form)."""
    _require_gpu(u, "u")
    if not delta_softplus:
        raise NotImplementedError("the Caduceus path always uses delta_softplus=True")
    if return_last_state:
        raise NotImplementedError("return_last_state is not on the Caduceus path")
    lib = load_library()
    Bsz, E, L = u.shape
    u_tm, z_tm, bc, A32, Dv, db = _scan_common(u, B, C, A, D, z, delta_bias)
    d_tm = delta.to(u.dtype).transpose(1, 2).contiguous()
    y = accumulate_into.transpose(1, 2).contiguous() if accumulate_into is not None else torch.empty_like(u_tm)
    with torch.cuda.device(u.device):
        _check(lib.pcad_selective_scan(u_tm.data_ptr(), d_tm.data_ptr(), z_tm.data_ptr() if z_tm is not None else None,
                                       E, bc.data_ptr(), A32.data_ptr(), Dv.data_ptr(), db.data_ptr(), y.data_ptr(),
                                       Bsz, L, E, int(bool(reverse)), _acc_mode(accumulate_into, gate_sum), _dt(u_tm),
                                       _stream_ptr()), "pcad_selective_scan")
    return y.transpose(1, 2)


def selective_scan_dtproj_fn(u, dt_low, dt_proj_weight, A, B, C, D=None, z=None, delta_bias=None, reverse=False,
                             accumulate_into=None, gate_sum=False):
    """mamba_inner_fn's tail: delta = dt_proj_weight @ dt_low (on MFMA inside the kernel), then selective_scan_fn.
    u, z: (B, E, L); dt_low: (B, L, R); dt_proj_weight: (E, R)."""
    _require_gpu(u, "u")
    lib = load_library()
    Bsz, E, L = u.shape
    R = dt_low.shape[-1]
    Rp = 64 if R <= 64 else (R + 31) // 32 * 32          # kernels.hpp padded_dt_rank
    u_tm, z_tm, bc, A32, Dv, db = _scan_common(u, B, C, A, D, z, delta_bias)
    dl = torch.zeros((Bsz, L, Rp), dtype=u.dtype, device=u.device)
    dl[..., :R] = dt_low.to(u.dtype)
    W = torch.zeros((E, Rp), dtype=u.dtype, device=u.device)
    W[:, :R] = dt_proj_weight.to(u.dtype)
    y = accumulate_into.transpose(1, 2).contiguous() if accumulate_into is not None else torch.empty_like(u_tm)
    with torch.cuda.device(u.device):
        _check(lib.pcad_selective_scan_dtproj(u_tm.data_ptr(), dl.data_ptr(), Rp, W.data_ptr(), Rp,
                                              z_tm.data_ptr() if z_tm is not None else None, E, bc.data_ptr(),
                                              A32.data_ptr(), Dv.data_ptr(), db.data_ptr(), y.data_ptr(), Bsz, L, E,
                                              int(bool(reverse)), _acc_mode(accumulate_into, gate_sum), _dt(u_tm),
                                              _stream_ptr()), "pcad_selective_scan_dtproj")
    return y.transpose(1, 2)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True,
                   reverse=False):
    """`mamba_ssm.ops.selective_scan_interface.mamba_inner_fn` (mamba-ssm 2.2.2, reference env/requirements.txt:10) - the ONE call
    `Mamba.forward`'s fast path makes after in_proj - under its own signature, on the engine's kernels:

        x, z    = xz.chunk(2, dim=1)                                        xz (B, 2E, L) channels-first, as upstream
        xc      = causal_conv1d_fn(x, conv1d_weight, conv1d_bias, "silu")   \  pcad_conv_xproj_bidir (one kernel; dt_rank <= 96),
        x_dbl   = F.linear(xc^T, x_proj_weight)          (R + 2N columns)   /  else pcad_causal_conv1d_silu + pcad_gemm_nt
        delta   = delta_proj_weight @ x_dbl[:, :R]^T                        \  pcad_selective_scan_dtproj (dt_proj on MFMA inside
        y       = selective_scan_fn(xc, delta, A, B_t, C_t, D, z, delta_bias, delta_softplus=True)   /  the scan; delta never stored)
        return    F.linear(y^T, out_proj_weight)                               pcad_gemm_nt            -> (B, L, D)

    conv1d_weight (E, 1, 4) or (E, 4); x_proj_weight (R + 32, E); delta_proj_weight (E, R); out_proj_weight (D, E);
    A (E, 16) = -exp(A_log) as upstream passes it; D, delta_bias (E).  B / C must be None (input-dependent, taken from x_dbl - the only
    form Mamba.forward uses), likewise the projection biases and out_proj_bias (Caduceus builds Mamba with bias=False).
    reverse=True is the twin BiMambaWrapper needs for `mamba_rev`: the same operator walking right-to-left with the anti-causal
    conv on the SAME rows, i.e. mamba_inner_fn(xz, ..., reverse=True) == mamba_inner_fn(xz.flip(-1), ...).flip(1) without either
    flip copy.  Tensors are ROCm tensors in one dtype (fp32 or bf16); parameters are converted as the engine's
    weight packing does (conv taps, A, D, biases to fp32; projections to xz.dtype)."""
    _require_gpu(xz, "xz")
    if B is not None or C is not None:
        raise NotImplementedError("constant B / C are not on the Caduceus path (Mamba.forward passes None: input-dependent B, C)")
    if out_proj_bias is not None or B_proj_bias is not None or C_proj_bias is not None:
        raise NotImplementedError("projection biases are not on the Caduceus path (Mamba(bias=False))")
    if not delta_softplus:
        raise NotImplementedError("the Caduceus path always uses delta_softplus=True")
    lib = load_library()
    Bsz, E2, L = xz.shape
    E = E2 // 2
    dt, dev = xz.dtype, xz.device
    if A.shape != (E, 16):
        raise ValueError(f"A must be (d_inner, 16), got {tuple(A.shape)}")
    R = delta_proj_weight.shape[1]
    if x_proj_weight.shape != (R + 32, E):
        raise ValueError(f"x_proj_weight must be (dt_rank + 2 * 16, d_inner) = ({R + 32}, {E}), got {tuple(x_proj_weight.shape)}")
    w = conv1d_weight.reshape(E, -1)
    if w.shape[1] != 4:
        raise NotImplementedError("only conv width 4")
    bias = conv1d_bias if conv1d_bias is not None else torch.zeros(E, device=dev)
    x_tm = xz[:, :E].transpose(1, 2).contiguous()                    # the engine is token-major (test plumbing: it never transposes)
    z_tm = xz[:, E:].transpose(1, 2).contiguous()
    rows = Bsz * L
    d = 1 if reverse else 0
    fused = R <= 96 and (E * xz.element_size()) % 128 == 0 and (rows + 16) * E * xz.element_size() < 2 ** 32
    if fused:
        xc_b, dtl, bc, _, Rp = _conv_xproj_raw(x_tm, w, bias, w, bias, x_proj_weight, x_proj_weight)
        xc = from_blocked(xc_b[d], rows)
        dl, bcd = dtl[d], bc[d]
    else:
        yf, yr = causal_conv1d_bidir(x_tm, w, bias, w, bias)
        xc = (yr if reverse else yf).reshape(rows, E)
        x_dbl = linear(xc, x_proj_weight)                            # [rows, R + 32] in xz.dtype (rounded like upstream's F.linear)
        Rp = 64 if R <= 64 else (R + 31) // 32 * 32
        dl = torch.zeros((rows, Rp), dtype=dt, device=dev)
        dl[:, :R] = x_dbl[:, :R]
        bcd = x_dbl[:, R:].float().contiguous()
    Wdt = torch.zeros((E, Rp), dtype=dt, device=dev)
    Wdt[:, :R] = delta_proj_weight.to(dt)
    A32 = A.float().contiguous()
    Dv = (D.float() if D is not None else torch.zeros(E, device=dev)).contiguous()
    db = (delta_bias.float() if delta_bias is not None else torch.zeros(E, device=dev)).contiguous()
    y = torch.empty((rows, E), dtype=dt, device=dev)
    with torch.cuda.device(dev):
        _check(lib.pcad_selective_scan_dtproj(xc.data_ptr(), dl.data_ptr(), Rp, Wdt.data_ptr(), Rp, z_tm.data_ptr(), E,
                                              bcd.data_ptr(), A32.data_ptr(), Dv.data_ptr(), db.data_ptr(), y.data_ptr(),
                                              Bsz, L, E, int(bool(reverse)), 0, _dt(xz), _stream_ptr()), "pcad_selective_scan_dtproj")
    return linear(y.view(Bsz, L, E), out_proj_weight)


def linear(x, weight, out_dtype: Optional[torch.dtype] = None):
    """F.linear(x, weight) on MFMA: x [..., K], weight [N, K] (same dtype), K*elem a multiple of 128 bytes."""
    _require_gpu(x, "x")
    lib = load_library()
    K = x.shape[-1]
    N = weight.shape[0]
    xf = x.contiguous().view(-1, K)
    wf = weight.to(x.dtype).contiguous()
    od = out_dtype or x.dtype
    out = torch.empty((xf.shape[0], N), dtype=od, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.pcad_gemm_nt(xf.data_ptr(), K, wf.data_ptr(), K, out.data_ptr(), N, xf.shape[0], N, K, _dt(xf),
                                _DT[od], _stream_ptr()), "pcad_gemm_nt")
    return out.view(*x.shape[:-1], N)


def linear_split(x, weight):
    """F.linear(x, weight) of fp32 tensors as ONE split-bf16 GEMM on the bf16 matrix pipes (include/pcad.h pcad_gemm_nt_split):
    x [..., K], weight [N, K], K % 64 == 0 -> fp32 [..., N]."""
    _require_gpu(x, "x")
    lib = load_library()
    if x.dtype != torch.float32:
        raise ValueError("linear_split takes fp32 tensors")
    K = x.shape[-1]
    N = weight.shape[0]
    xf = x.contiguous().view(-1, K)
    wf = weight.float().contiguous()
    M = xf.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    nb = lib.pcad_gemm_nt_split_scratch_bytes(M, N, K)
    scratch = torch.empty(nb + 256, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.pcad_gemm_nt_split(xf.data_ptr(), K, wf.data_ptr(), K, out.data_ptr(), N, M, N, K,
                                      (scratch.data_ptr() + 255) // 256 * 256, nb, _stream_ptr()), "pcad_gemm_nt_split")
    return out.view(*x.shape[:-1], N)


def to_res_fragment(res: torch.Tensor) -> torch.Tensor:
    """[M, N] (M, N multiples of 256) -> the same values in the 4-wave GEMM's fragment layout (include/pcad.h
    pcad_gemm_nt_residual), as a flat [M * N] tensor."""
    M, N = res.shape
    if M % 256 or N % 256:
        raise ValueError("the fragment layout needs M % 256 == 0 and N % 256 == 0")
    #          tm        wm i  li   tn        wn jg lg k  r
    v = res.reshape(M // 256, 2, 8, 16, N // 256, 2, 2, 4, 4, 4)
    #  -> [tm, tn, wm, wn, i, jg, k, lg, li, r]
    return v.permute(0, 4, 1, 5, 2, 6, 8, 7, 3, 9).contiguous().view(-1)


def from_res_fragment(frag: torch.Tensor, M: int, N: int) -> torch.Tensor:
    v = frag.view(M // 256, N // 256, 2, 2, 8, 2, 4, 4, 16, 4)
    #  [tm, tn, wm, wn, i, jg, k, lg, li, r] -> [tm, wm, i, li, tn, wn, jg, lg, k, r]
    return v.permute(0, 2, 4, 8, 1, 3, 5, 7, 6, 9).contiguous().view(M, N)


def linear_residual(x, weight, residual):
    """The "norm_fold" out_proj as an operator (include/pcad.h pcad_gemm_nt_residual): residual fp32 [M, N] + x [M, K] @ weight
    [N, K]^T.  Returns (new residual fp32 [M, N], round(new residual) in x.dtype [M, N], ssq [M, N/128] per-row partial sums of
    squares).  The kernel updates the residual in place in its fragment layout; the conversions are done here (test plumbing)."""
    _require_gpu(x, "x")
    lib = load_library()
    M, K = x.shape
    N = weight.shape[0]
    if M % 256 or N % 256:
        raise RuntimeError("pcad_gemm_nt_residual: M and N must be multiples of 256")
    if residual.dtype != torch.float32 or tuple(residual.shape) != (M, N):
        raise ValueError("residual must be an fp32 [M, N] tensor")
    xf = x.contiguous()
    wf = weight.to(x.dtype).contiguous()
    frag = to_res_fragment(residual)
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    ssq = torch.empty((M, N // 128), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.pcad_gemm_nt_residual(xf.data_ptr(), K, wf.data_ptr(), K, out.data_ptr(), frag.data_ptr(), ssq.data_ptr(),
                                         M, N, K, _dt(xf), _stream_ptr()), "pcad_gemm_nt_residual")
    return from_res_fragment(frag, M, N), out, ssq


def gather_rows(src, B: int, L: int, positions):
    """src [2B*L, E] -> [2B*P, E]: strand b row p_q, strand B+b row L-1-p_q (include/pcad.h pcad_gather_rows)."""
    import ctypes as C
    _require_gpu(src, "src")
    lib = load_library()
    E = src.shape[-1]
    P = len(positions)
    srcf = src.contiguous()
    out = torch.empty((2 * B * P, E), dtype=src.dtype, device=src.device)
    arr = (C.c_int32 * max(P, 1))(*[int(p) for p in positions])
    with torch.cuda.device(src.device):
        _check(lib.pcad_gather_rows(srcf.data_ptr(), out.data_ptr(), B, L, E, arr, P, _dt(srcf), _stream_ptr()), "pcad_gather_rows")
    return out


def final_head(h, res, norm_weight, emb, complement, B: int, L: int, eps: float, positions=None, pos_per_seq=None,
               h_compact: bool = False, want_hidden: bool = True, want_logits: bool = True, ids=None, status=None,
               res_fragment: bool = False):
    """norm_f + RC re-assembly + tied RCPS LM head at the requested positions (include/pcad.h pcad_final_head).
    h [2B*L, D] (or the gathered rows when h_compact), res [2B*L, D] (fp32 or h.dtype), emb [V, D] (rounded to h.dtype here, as the
    tied lm_head weight is).  -> (hidden [B, Q, 2D] | None, logits fp32 [B, Q, V] | None)."""
    import ctypes as C
    _require_gpu(h, "h")
    lib = load_library()
    D = h.shape[-1]
    V = emb.shape[0]
    P = 0 if positions is None else len(positions)
    Q = 1 if pos_per_seq is not None else (P if P else L)
    hf, rf = h.contiguous(), res.contiguous()
    w = norm_weight.float().contiguous()
    e32 = emb.to(h.dtype).float().contiguous()
    comp = torch.as_tensor(list(complement), dtype=torch.int32, device=h.device)
    hid = torch.empty((B, Q, 2 * D), dtype=h.dtype, device=h.device) if want_hidden else None
    lg = torch.empty((B, Q, V), dtype=torch.float32, device=h.device) if want_logits else None
    arr = (C.c_int32 * max(P, 1))(*[int(p) for p in (positions or [])]) if P else None
    pps = pos_per_seq.to(torch.int32).contiguous() if pos_per_seq is not None else None
    with torch.cuda.device(h.device):
        _check(lib.pcad_final_head(hf.data_ptr(), rf.data_ptr(), w.data_ptr(), e32.data_ptr(), comp.data_ptr(),
                                   hid.data_ptr() if hid is not None else None, lg.data_ptr() if lg is not None else None,
                                   B, L, D, float(eps), arr, P, pps.data_ptr() if pps is not None else None, int(bool(h_compact)),
                                   ids.data_ptr() if ids is not None else None, status.data_ptr() if status is not None else None,
                                   _dt(hf), _DT[rf.dtype], int(bool(res_fragment)), _stream_ptr()), "pcad_final_head")
    return hid, lg
