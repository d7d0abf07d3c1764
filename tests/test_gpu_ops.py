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
                                           (3, 77, 768, 24, torch.bfloat16), (2, 133, 1536, 48, torch.float32),
                                           # dt_rank 65..96 (PlantCAD2 Large: d_inner 3072, dt_rank 96): the 8-fragment variant, one Wx slab
                                           (2, 512, 3072, 96, torch.bfloat16), (3, 200, 512, 80, torch.float32),
                                           (2, 77, 1024, 96, torch.bfloat16), (1, 384, 3072, 96, torch.float32), (2, 5, 256, 65, torch.float32)])
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
                                       (256, 64, torch.bfloat16), (45, 24, torch.bfloat16),
                                       # dt_rank 65..96: K of the in-kernel dt_proj padded to 96 (PlantCAD2 Large), up to 128 generically
                                       (512, 96, torch.bfloat16), (100, 80, torch.float32), (64, 96, torch.float32), (39, 80, torch.bfloat16),
                                       (48, 128, torch.float32)])
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


# ---- "norm_fold" out_proj (include/pcad.h pcad_gemm_nt_residual): res += x . W^T, round(res), per-slab sums of squares ----------
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 1024, 2048), (2304, 768, 192), (66048, 512, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_residual(ops, M, N, K, dtype):
    """one to 258 tiles per workgroup; the accumulators start as the residual tile (loads straight into the accumulator
    registers, waited for row by row in the first k-step), so every element checks that hand-counted wait."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    res = torch.randn(M, N, generator=g) * 3
    ref = res.double() + x.double() @ w.double().t()
    r_dev, out, ssq = ops.linear_residual(x.to(DEV), w.to(DEV), res.to(DEV))
    tol = 5e-6 if dtype == torch.float32 else 1e-5          # fp32: exact products, only the accumulation order differs (K up to 2048, |res| ~ 3 sigma 9); bf16 products are exact in fp32
    assert relerr(r_dev, ref) < tol
    assert torch.equal(out.cpu(), r_dev.cpu().to(dtype))                 # C is the updated residual rounded once
    want = (r_dev.double().cpu() ** 2).view(M, N // 128, 128).sum(-1)
    assert ssq.shape == (M, N // 128) and relerr(ssq, want) < 1e-5
    # deterministic: a second run on the same inputs gives the same bits
    r2, out2, ssq2 = ops.linear_residual(x.to(DEV), w.to(DEV), res.to(DEV))
    assert torch.equal(r2, r_dev) and torch.equal(ssq2, ssq) and torch.equal(out2, out)


def test_linear_residual_rejects_partial_tiles(ops):
    x = torch.zeros(300, 64, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 256"):
        ops.linear_residual(x, torch.zeros(256, 64, device=DEV), torch.zeros(300, 256, device=DEV))


# ---- the forward's last two kernels as operators ---------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,E,pos", [(3, 40, 256, [0]), (2, 64, 512, [63, 0, 63, 17]), (5, 33, 128, [32, 1])])
def test_gather_rows(ops, dtype, B, L, E, pos):
    """positions at 0, at L - 1, duplicated: strand b row p and strand B + b row L - 1 - p, in (strand, q) order; bit copies."""
    g = torch.Generator().manual_seed(B * L + E)
    src = torch.randn(2 * B * L, E, generator=g).to(dtype)
    got = ops.gather_rows(src.to(DEV), B, L, pos).cpu()
    rows = [s * L + (p if s < B else L - 1 - p) for s in range(2 * B) for p in pos]
    assert torch.equal(got, src[rows])


def _final_head_ref(h, res, w, emb, comp, B, L, eps, rows_p, dtype):
    """res + h -> norm_f (rounded to dtype) -> cat(fwd, reverse_channels(rc)) and tied RCPS head at positions rows_p[b] (list)"""
    rnd = (lambda t: t.to(dtype).float()) if dtype != torch.float32 else (lambda t: t)
    D = h.shape[-1]
    s = h.float() + res.float()
    H = rnd(s * torch.rsqrt((s * s).mean(-1, keepdim=True) + eps) * w.float())
    e32 = rnd(emb.float())
    hid, lg = [], []
    for b in range(B):
        hb, lb = [], []
        for p in rows_p[b]:
            f, r = H[b * L + p], H[(B + b) * L + (L - 1 - p)]
            hb.append(torch.cat([f, r.flip(0)]))
            lb.append(rnd(rnd(e32 @ f) + rnd(e32[comp] @ r)))
        hid.append(torch.stack(hb)); lg.append(torch.stack(lb))
    return torch.stack(hid), torch.stack(lg)


@pytest.mark.parametrize("dtype,rdtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("B,L,D", [(3, 24, 64), (2, 50, 384), (2, 16, 1024), (1, 9, 2048), (2, 64, 512)])
def test_final_head(ops, dtype, rdtype, B, L, D):
    g = torch.Generator().manual_seed(B + L + D)
    h = torch.randn(2 * B * L, D, generator=g).to(dtype)
    res = (torch.randn(2 * B * L, D, generator=g) * 2).to(rdtype)
    w = torch.rand(D, generator=g) + 0.5
    emb = torch.randn(8, D, generator=g) * 0.1
    comp = [0, 1, 2, 6, 5, 4, 3, 7]
    tol_h, tol_l = (1e-5, 2e-5) if dtype == torch.float32 else (2 ** -7, 2 ** -6)
    # shared positions incl. 0, L - 1 and a duplicate; all positions; one position per window
    pos = [0, L - 1, L // 2, 0]
    for kw, rows_p in ((dict(positions=pos), [pos] * B), (dict(), [list(range(L))] * B),
                       (dict(pos_per_seq=torch.tensor([(7 * b) % L for b in range(B)], dtype=torch.int32, device=DEV)),
                        [[(7 * b) % L] for b in range(B)])):
        hid, lg = ops.final_head(h.to(DEV), res.to(DEV), w.to(DEV), emb.to(DEV), comp, B, L, 1e-5, **kw)
        hid_ref, lg_ref = _final_head_ref(h, res, w, emb, comp, B, L, 1e-5, rows_p, dtype)
        assert relerr(hid, hid_ref) < tol_h and relerr(lg, lg_ref) < tol_l
    # compact h (the last-layer shortcut's gathered rows) == the full tensor, bit for bit, with everything around the
    # consumed rows poisoned
    hc = ops.gather_rows(h.to(DEV), B, L, pos)
    hid_c, lg_c = ops.final_head(hc, res.to(DEV), w.to(DEV), emb.to(DEV), comp, B, L, 1e-5, positions=pos, h_compact=True)
    hid_f, lg_f = ops.final_head(h.to(DEV), res.to(DEV), w.to(DEV), emb.to(DEV), comp, B, L, 1e-5, positions=pos)
    assert torch.equal(hid_c, hid_f) and torch.equal(lg_c, lg_f)
    # the fp32 residual in the folded GEMM's fragment layout (norm_fold): same bits as from plain rows
    if rdtype == torch.float32 and D % 256 == 0 and (2 * B * L) % 256 == 0:
        frag = ops.to_res_fragment(res.to(DEV))
        assert torch.equal(ops.from_res_fragment(frag, 2 * B * L, D), res.to(DEV))
        hid_g, lg_g = ops.final_head(h.to(DEV), frag, w.to(DEV), emb.to(DEV), comp, B, L, 1e-5, positions=pos, res_fragment=True)
        assert torch.equal(hid_g, hid_f) and torch.equal(lg_g, lg_f)


def test_final_head_clamps_and_flags_bad_positions(ops):
    """a per-window position outside [0, L) is clamped (nothing is read out of bounds: the tensors sit at the END of their
    allocations' used range, surrounded by NaN guard rows) and reported through the status word; bad token ids likewise."""
    B, L, D = 3, 12, 128
    guard = 4 * L
    big_h = torch.full((2 * B * L + 2 * guard, D), float("nan"), device=DEV)
    big_r = torch.full((2 * B * L + 2 * guard, D), float("nan"), device=DEV)
    h, res = big_h[guard:guard + 2 * B * L], big_r[guard:guard + 2 * B * L]
    h.normal_(); res.normal_()
    w, emb = torch.ones(D, device=DEV), torch.randn(8, D, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ids = torch.randint(3, 7, (B, L), dtype=torch.int32, device=DEV)
    pps = torch.tensor([5, L + 100, -3], dtype=torch.int32, device=DEV)
    hid, lg = ops.final_head(h, res, w, emb, [0, 1, 2, 6, 5, 4, 3, 7], B, L, 1e-5, pos_per_seq=pps, ids=ids, status=status)
    assert int(status.item()) == 2 and torch.isfinite(hid).all() and torch.isfinite(lg).all()
    ok, _ = ops.final_head(h, res, w, emb, [0, 1, 2, 6, 5, 4, 3, 7], B, L, 1e-5,
                           pos_per_seq=torch.tensor([5, L - 1, 0], dtype=torch.int32, device=DEV))
    assert torch.equal(hid, ok)                                  # clamped to L - 1 and 0
    status.zero_(); ids[1, 3] = 9
    ops.final_head(h, res, w, emb, [0, 1, 2, 6, 5, 4, 3, 7], B, L, 1e-5, positions=[1], ids=ids, status=status)
    assert int(status.item()) == 1


def _mamba_params(seed, D, E, R, scale=1.0):
    """One Mamba's parameters with the init ranges of plantcaduceus_amd.checkpoint.synthetic_state_dict (distinct per seed)."""
    import math
    g = torch.Generator().manual_seed(seed)

    def U(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    dt = torch.exp(torch.rand(E, generator=g) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3)).clamp_min(1e-4)
    return O.MambaParams(
        in_proj=U((2 * E, D), D ** -0.5), conv_w=U((E, 4), 0.5), conv_b=U((E,), 0.5), x_proj=U((R + 32, E), E ** -0.5) * scale,
        dt_proj_w=U((E, R), R ** -0.5), dt_proj_b=dt + torch.log(-torch.expm1(-dt)),
        A_log=torch.log(torch.arange(1, 17, dtype=torch.float32).repeat(E, 1)) + 0.3 * torch.randn(E, 16, generator=g),
        D=torch.rand(E, generator=g) + 0.5, out_proj=U((D, E), E ** -0.5))


@pytest.mark.parametrize("D,R,Bsz,L", [(384, 24, 2, 512), (1024, 64, 2, 512), (384, 24, 3, 77), (128, 8, 2, 40), (768, 128, 1, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mamba_inner_fn(ops, D, R, Bsz, L, dtype):
    """ops.mamba_inner_fn under mamba-ssm 2.2.2's signature (xz channels-first (B, 2E, L) in, (B, L, D) out) against the oracle's
    restatement of the same function, at the BASELINE widths (D = 384: l20, 1024: l32) and off them (ragged L, dt_rank 128: the
    unfused conv / x_proj pair); fp32 <= 3e-5 of the output's max, bf16 <= 2^-7 against the oracle rounding to bf16 where
    upstream stores a bf16 tensor (conv out, x_dbl, delta, scan out, out_proj out).  And the reverse twin BiMambaWrapper needs:
    reverse=True on the unflipped rows == the plain operator on the flipped sequence, flipped back."""
    E = 2 * D
    p = _mamba_params(D + R, D, E, R)
    g = torch.Generator().manual_seed(L)
    xz = torch.randn(Bsz, 2 * E, L, generator=g)
    rnd = O.round_bf16 if dtype == torch.bfloat16 else (lambda t: t)
    if dtype == torch.bfloat16:                        # from_pretrained(torch_dtype=bf16): parameters and the input are bf16 tensors
        p = O.MambaParams(**{k: v.bfloat16().float() for k, v in vars(p).items()})
        xz = xz.bfloat16().float()
    tol = 2 ** -7 if dtype == torch.bfloat16 else 3e-5
    A = -torch.exp(p.A_log)
    args = [t.to(DEV) for t in (p.conv_w.unsqueeze(1), p.conv_b, p.x_proj.to(dtype), p.dt_proj_w.to(dtype), p.out_proj.to(dtype))]
    kw = dict(D=p.D.to(DEV), delta_bias=p.dt_proj_b.to(DEV), delta_softplus=True)
    out = ops.mamba_inner_fn(xz.to(dtype).to(DEV), args[0], args[1], args[2], args[3], args[4], None, A.to(DEV), None, None, **kw)
    ref = O.mamba_inner(xz, p, rnd)
    assert out.shape == (Bsz, L, D) and out.dtype == dtype
    e = relerr(out, ref)
    out_r = ops.mamba_inner_fn(xz.to(dtype).to(DEV), args[0], args[1], args[2], args[3], args[4], None, A.to(DEV), None, None,
                               reverse=True, **kw)
    ref_r = O.mamba_inner(xz.flip(-1), p, rnd).flip(1)
    er = relerr(out_r, ref_r)
    print(f"mamba_inner_fn D={D} R={R} L={L} {dtype}: rel err {e:.2e} (reverse twin {er:.2e})")
    assert e < tol and er < tol
    with pytest.raises(NotImplementedError):
        ops.mamba_inner_fn(xz.to(dtype).to(DEV), args[0], args[1], args[2], args[3], args[4], torch.zeros(D, device=DEV), A.to(DEV))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (2048, 4096, 1024), (4096, 1024, 2048), (300, 96, 768), (1000, 384, 768), (513, 1536, 384)])
def test_linear_split(ops, M, N, K):
    """pcad_gemm_nt_split (the fp32 model's in_proj / out_proj under "f32_gemm_split"): fp32 operands carried as two bf16 values,
    three bf16 MFMA products per fp32 product, fp32 accumulation.  Against a float64 product of the same fp32 inputs: <= 2e-5 of
    the result's max (operand error 2^-17 per element; measured ~2e-6), and at least 50x closer than the plain bf16 GEMM of the
    rounded operands; shapes on and off the 256 x 256 tiles (in_proj / out_proj of l32 and l20)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    ref = (x.double() @ w.double().t())
    out = ops.linear_split(x.to(DEV), w.to(DEV))
    assert out.dtype == torch.float32 and out.shape == (M, N)
    e = ((out.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    e32 = ((ops.linear(x.to(DEV), w.to(DEV)).double().cpu() - ref).abs().max() / ref.abs().max()).item()
    if K % 64 == 0 and (K * 2) % 128 == 0:
        ebf = ((ops.linear(x.bfloat16().to(DEV), w.bfloat16().to(DEV), out_dtype=torch.float32).double().cpu() - ref).abs().max() / ref.abs().max()).item()
    else:
        ebf = 1.0
    print(f"linear_split {M}x{N}x{K}: rel err {e:.2e} (fp32 MFMA GEMM {e32:.2e}, bf16 GEMM {ebf:.2e})")
    assert e < 2e-5 and e * 50 < ebf
