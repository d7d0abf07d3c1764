"""-m gpu: operator-level parity of each HIP kernel (through the C ABI) against the CPU oracle's
restatement of the operator it replaces, on the same seeded inputs.  fp32: <=1e-5 relative to the
tensor's max magnitude; bf16 I/O: bf16-rounding tolerance (2^-8 relative) stated per test."""
import pytest
import torch

from oracle import caduceus_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def relerr(a, b):
    return ((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from plantcaduceus_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("D", [64, 384, 1024, 2048])
@pytest.mark.parametrize("with_res", [False, True])
def test_rms_norm_fp32(ops, D, with_res):
    g = torch.Generator().manual_seed(D)
    x = torch.randn(37, D, generator=g)
    r = torch.randn(37, D, generator=g) if with_res else None
    w = torch.rand(D, generator=g) + 0.5
    y_ref, r_ref = O.rms_norm_fn(x, w, residual=r, eps=1e-5, prenorm=True, residual_in_fp32=True)
    y, ro = ops.rms_norm_fn(x.to(DEV), w.to(DEV), None, residual=None if r is None else r.to(DEV), eps=1e-5,
                            prenorm=True, residual_in_fp32=True)
    assert relerr(y, y_ref) < 1e-5
    assert relerr(ro, r_ref) < 1e-6


def test_rms_norm_bf16(ops):
    g = torch.Generator().manual_seed(1)
    D = 1024
    x = torch.randn(50, D, generator=g).bfloat16()
    r = torch.randn(50, D, generator=g)
    w = torch.rand(D, generator=g) + 0.5
    y_ref, r_ref = O.rms_norm_fn(x.float(), w, residual=r, eps=1e-5, prenorm=True, residual_in_fp32=True,
                                 rnd=O.round_bf16)
    y, ro = ops.rms_norm_fn(x.to(DEV), w.to(DEV), None, residual=r.to(DEV), eps=1e-5, prenorm=True,
                            residual_in_fp32=True)
    assert y.dtype == torch.bfloat16 and ro.dtype == torch.float32
    assert relerr(ro, r_ref) < 1e-6
    assert relerr(y, y_ref) < 2 ** -7   # one bf16 ulp of slack on a bf16 output


@pytest.mark.parametrize("L", [1, 3, 5, 24, 512])
def test_causal_conv1d_fp32(ops, L):
    g = torch.Generator().manual_seed(L)
    Bsz, E = 3, 128
    x = torch.randn(Bsz, E, L, generator=g)
    w = torch.randn(E, 4, generator=g) * 0.5
    b = torch.randn(E, generator=g) * 0.5
    ref = O.causal_conv1d_fn(x, w, b, activation="silu")
    out = ops.causal_conv1d_fn(x.to(DEV), w.to(DEV), b.to(DEV), activation="silu")
    assert relerr(out, ref) < 1e-5
    # anti-causal direction == causal conv of the flipped sequence, flipped back
    w2 = torch.randn(E, 4, generator=g) * 0.5
    b2 = torch.randn(E, generator=g) * 0.5
    ref_rev = O.causal_conv1d_fn(x.flip(-1), w2, b2, activation="silu").flip(-1)
    yf, yr = ops.causal_conv1d_bidir(x.transpose(1, 2).contiguous().to(DEV), w.to(DEV), b.to(DEV), w2.to(DEV), b2.to(DEV))
    assert relerr(yf.transpose(1, 2), ref) < 1e-5
    assert relerr(yr.transpose(1, 2), ref_rev) < 1e-5


@pytest.mark.parametrize("S,L,E,R,dtype", [(3, 24, 128, 8, torch.float32), (2, 512, 768, 24, torch.float32),
                                           (2, 203, 256, 16, torch.float32), (1, 1, 128, 8, torch.float32),
                                           (2, 5, 128, 8, torch.float32), (4, 512, 2048, 64, torch.bfloat16),
                                           (3, 77, 768, 24, torch.bfloat16), (2, 133, 1536, 48, torch.float32)])
def test_conv_xproj_fused(ops, S, L, E, R, dtype):
    """pcad_conv_xproj_bidir — the kernel the engine actually runs for conv1d+SiLU and x_proj (convx.hip): both directions
    against causal_conv1d_fn + einsum of the oracle (mamba_inner's head).  fp32: 1e-5 (conv) / 3e-5 (x_dbl, K = E sums);
    bf16: one bf16 ulp on xc, x_dbl vs the oracle fed the kernel's own bf16 xc (isolates the GEMM) 2^-7."""
    g = torch.Generator().manual_seed(S * 1000 + L)
    x = torch.randn(S, L, E, generator=g)
    wf, wr = (torch.randn(E, 4, generator=g) * 0.5 for _ in range(2))
    bf, br = (torch.randn(E, generator=g) * 0.5 for _ in range(2))
    xpf, xpr = (torch.randn(R + 32, E, generator=g) * E ** -0.5 for _ in range(2))
    if dtype == torch.bfloat16:
        x, xpf, xpr = (O.round_bf16(t) for t in (x, xpf, xpr))
    rnd = O.round_bf16 if dtype == torch.bfloat16 else O._ident
    xcf_ref = O.causal_conv1d_fn(x.transpose(1, 2), wf, bf, activation="silu", rnd=rnd).transpose(1, 2)
    xcr_ref = O.causal_conv1d_fn(x.transpose(1, 2).flip(-1), wr, br, activation="silu", rnd=rnd).flip(-1).transpose(1, 2)
    xcf, xcr, dblf, dblr = ops.conv_xproj_bidir(x.to(dtype).to(DEV), wf.to(DEV), bf.to(DEV), wr.to(DEV), br.to(DEV),
                                                xpf.to(dtype).to(DEV), xpr.to(dtype).to(DEV))
    assert xcf.dtype == dtype and dblf.shape == (S, L, R + 32)
    tol_c = 1e-5 if dtype == torch.float32 else 2 ** -7
    assert relerr(xcf, xcf_ref) < tol_c and relerr(xcr, xcr_ref) < tol_c
    for xc, dbl, w in ((xcf, dblf, xpf), (xcr, dblr, xpr)):
        ref = rnd(torch.einsum("sle,re->slr", xc.float().cpu(), w))          # the GEMM on the kernel's own conv output
        assert relerr(dbl, ref) < (3e-5 if dtype == torch.float32 else 2 ** -7)
    if dtype == torch.float32:                                               # end to end against the oracle's conv output
        assert relerr(dblf, torch.einsum("sle,re->slr", xcf_ref, xpf)) < 3e-5
        assert relerr(dblr, torch.einsum("sle,re->slr", xcr_ref, xpr)) < 3e-5


def _scan_inputs(seed, Bsz, E, L, N=16):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(Bsz, E, L, generator=g)
    delta = torch.randn(Bsz, E, L, generator=g) * 0.5 - 3.0
    A = -torch.exp(torch.log(torch.arange(1, N + 1).float())[None, :] + 0.3 * torch.randn(E, N, generator=g))
    Bm = torch.randn(Bsz, N, L, generator=g)
    Cm = torch.randn(Bsz, N, L, generator=g)
    D = torch.rand(E, generator=g) + 0.5
    z = torch.randn(Bsz, E, L, generator=g)
    db = torch.randn(E, generator=g)
    return u, delta, A, Bm, Cm, D, z, db


@pytest.mark.parametrize("L", [1, 7, 8, 9, 64, 512])
def test_selective_scan_fp32(ops, L):
    u, delta, A, Bm, Cm, D, z, db = _scan_inputs(L, 2, 128, L)
    ref = O.selective_scan_fn(u, delta, A, Bm, Cm, D, z=z, delta_bias=db, delta_softplus=True)
    to = lambda t: t.to(DEV)
    out = ops.selective_scan_fn(to(u), to(delta), to(A), to(Bm), to(Cm), to(D), z=to(z), delta_bias=to(db),
                                delta_softplus=True)
    assert relerr(out, ref) < 2e-5
    # reverse direction + accumulate == fwd(x) + flip(fwd(flip(x)))
    ref_rev = O.selective_scan_fn(u.flip(-1), delta.flip(-1), A, Bm.flip(-1), Cm.flip(-1), D, z=z.flip(-1),
                                  delta_bias=db, delta_softplus=True).flip(-1)
    out2 = ops.selective_scan_fn(to(u), to(delta), to(A), to(Bm), to(Cm), to(D), z=to(z), delta_bias=to(db),
                                 delta_softplus=True, reverse=True, accumulate_into=out)
    assert relerr(out2, ref + ref_rev) < 2e-5
    # the engine's form: forward ungated, reverse adds and gates the sum once: (y_f + y_r) * silu(z)
    raw_f = O.selective_scan_fn(u, delta, A, Bm, Cm, D, z=None, delta_bias=db, delta_softplus=True)
    raw_r = O.selective_scan_fn(u.flip(-1), delta.flip(-1), A, Bm.flip(-1), Cm.flip(-1), D, z=None, delta_bias=db,
                                delta_softplus=True).flip(-1)
    o_f = ops.selective_scan_fn(to(u), to(delta), to(A), to(Bm), to(Cm), to(D), z=None, delta_bias=to(db),
                                delta_softplus=True)
    assert relerr(o_f, raw_f) < 2e-5
    o_s = ops.selective_scan_fn(to(u), to(delta), to(A), to(Bm), to(Cm), to(D), z=to(z), delta_bias=to(db),
                                delta_softplus=True, reverse=True, accumulate_into=o_f, gate_sum=True)
    assert relerr(o_s, (raw_f + raw_r) * torch.nn.functional.silu(z)) < 2e-5
    assert relerr(o_s, ref + ref_rev) < 2e-5            # == y_f*g + y_r*g up to fp32 rounding


@pytest.mark.parametrize("L,R,dtype", [(512, 64, torch.float32), (70, 24, torch.float32), (33, 48, torch.float32),
                                       (256, 64, torch.bfloat16), (45, 24, torch.bfloat16)])
def test_selective_scan_fused_dtproj(ops, L, R, dtype):
    """engine form: delta = dt_proj.weight @ dt_low on MFMA inside the scan (mamba_inner_fn tail), both directions."""
    u, _, A, Bm, Cm, D, z, db = _scan_inputs(L + R, 2, 128, L)
    g = torch.Generator().manual_seed(R)
    dt_low = torch.randn(2, L, R, generator=g)
    Wdt = torch.randn(128, R, generator=g) * R ** -0.5
    db = db - 3.0
    bf = dtype == torch.bfloat16
    rnd = O.round_bf16 if bf else (lambda t: t)
    c = (lambda t: t.bfloat16().float()) if bf else (lambda t: t)
    delta = rnd(torch.einsum("blr,er->bel", c(dt_low), c(Wdt)))
    ref = O.selective_scan_fn(c(u), delta, A, c(Bm), c(Cm), D, z=c(z), delta_bias=db, delta_softplus=True, rnd=rnd)
    ref_rev = O.selective_scan_fn(c(u).flip(-1), delta.flip(-1), A, c(Bm).flip(-1), c(Cm).flip(-1), D, z=c(z).flip(-1),
                                  delta_bias=db, delta_softplus=True, rnd=rnd).flip(-1)
    to = lambda t: t.to(DEV).to(dtype) if t.is_floating_point() else t.to(DEV)
    f = lambda t: t.to(DEV)
    out = ops.selective_scan_dtproj_fn(to(u), to(dt_low), to(Wdt), f(A), to(Bm), to(Cm), f(D), z=to(z), delta_bias=f(db))
    tol = 2 ** -7 if bf else 3e-5
    assert out.dtype == dtype
    assert relerr(out, ref) < tol
    out2 = ops.selective_scan_dtproj_fn(to(u), to(dt_low), to(Wdt), f(A), to(Bm), to(Cm), f(D), z=to(z), delta_bias=f(db),
                                        reverse=True, accumulate_into=out)
    assert relerr(out2, rnd(ref + ref_rev)) < 2 * tol
    o_f = ops.selective_scan_dtproj_fn(to(u), to(dt_low), to(Wdt), f(A), to(Bm), to(Cm), f(D), z=None, delta_bias=f(db))
    o_s = ops.selective_scan_dtproj_fn(to(u), to(dt_low), to(Wdt), f(A), to(Bm), to(Cm), f(D), z=to(z), delta_bias=f(db),
                                       reverse=True, accumulate_into=o_f, gate_sum=True)
    assert relerr(o_s, rnd(ref + ref_rev)) < 2 * tol    # gate-once form: a rounding-order difference only


def test_selective_scan_softplus_threshold_and_small_dt(ops):
    """edge cases: softplus linear branch (>20) and tiny time-steps (log1p accuracy)."""
    u, delta, A, Bm, Cm, D, z, db = _scan_inputs(3, 1, 64, 32)
    delta[:, :16] = 25.0
    delta[:, 16:32] = -12.0
    db.zero_()
    A = A * 0.01
    ref = O.selective_scan_fn(u, delta, A, Bm, Cm, D, z=z, delta_bias=db, delta_softplus=True)
    to = lambda t: t.to(DEV)
    out = ops.selective_scan_fn(to(u), to(delta), to(A), to(Bm), to(Cm), to(D), z=to(z), delta_bias=to(db),
                                delta_softplus=True)
    err = (out.cpu() - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())
    assert err.max().item() < 1e-4


def test_selective_scan_bf16(ops):
    u, delta, A, Bm, Cm, D, z, db = _scan_inputs(11, 2, 128, 256)
    bf = lambda t: t.bfloat16()
    ref = O.selective_scan_fn(bf(u).float(), bf(delta).float(), A, bf(Bm).float(), bf(Cm).float(), D, z=bf(z).float(),
                              delta_bias=db, delta_softplus=True, rnd=O.round_bf16)
    to = lambda t: t.to(DEV)
    out = ops.selective_scan_fn(to(bf(u)), to(bf(delta)), to(A), to(bf(Bm)), to(bf(Cm)), to(D), z=to(bf(z)),
                                delta_bias=to(db), delta_softplus=True)
    assert out.dtype == torch.bfloat16
    assert relerr(out, ref) < 2 ** -7


# (2048, 512, 96), (2304, 768, 32): full 256x256 tiles -> the 4-wave kernel with 3 / 1 K-tiles; (2100, 520, 64): its 8-wave edge-tile sibling
@pytest.mark.parametrize("M,N,K", [(1, 8, 32), (128, 128, 32), (300, 96, 768), (1000, 1536, 384), (257, 56, 2048), (513, 384, 64),
                                   (2048, 512, 96), (2304, 768, 32), (2100, 520, 64)])
def test_linear_fp32(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    ref = (x.double() @ w.double().t()).float()
    out = ops.linear(x.to(DEV), w.to(DEV))
    assert relerr(out, ref) < 2e-6      # exact-fp32 MFMA: only summation-order differences


# (2048, 4096, 1024), (2304, 768, 64), (2560, 512, 192): 4-wave 256x256 kernel (16 / 1 / 3 K-tiles, 1..9 tiles per block);
# (2100, 1000, 128): the 8-wave ring kernel with edge tiles in M and N
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 96, 768), (1000, 1536, 384), (2048, 4096, 1024), (513, 1024, 2048),
                                   (2304, 768, 64), (2560, 512, 192), (2100, 1000, 128), (65536, 512, 128)])
def test_linear_bf16(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    ref = x.double() @ w.double().t()
    out32 = ops.linear(x.to(DEV), w.to(DEV), out_dtype=torch.float32)
    assert relerr(out32, ref) < 1e-5    # bf16 products are exact in fp32; fp32 accumulation
    out = ops.linear(x.to(DEV), w.to(DEV))
    assert out.dtype == torch.bfloat16
    assert torch.equal(out.cpu(), out32.cpu().bfloat16()) or relerr(out, ref) < 2 ** -8
