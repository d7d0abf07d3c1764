"""-m gpu: model-level parity of the HIP forward (through the C ABI + HF surface) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import caduceus_oracle as O
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, sd, dtype, **engine_options):
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    cfg.engine_options = dict(engine_options)        # pcad_set_option at engine creation
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    return m.to(dtype).to(DEV)


def rand_ids(B, L, seed, mask=None):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 7, (B, L), generator=g)
    ids[0, 0] = 2   # an N -> [UNK]
    if mask is not None:
        ids[:, mask] = 1
    return ids


@pytest.mark.parametrize("D,nl,B,L", [(64, 3, 2, 24), (128, 2, 5, 64), (128, 2, 3, 45), (384, 2, 2, 203), (384, 2, 3, 512),
                                      (1024, 1, 2, 512)])
def test_forward_fp32_matches_oracle(D, nl, B, L):
    """north_star tolerance: <=1e-4 relative on fp32 logits, exact argmax of the nucleotide call."""
    cfg = make_config("x", d_model=D, n_layer=nl)
    sd = synthetic_state_dict(cfg, seed=D + nl)
    ids = rand_ids(B, L, 5, mask=L // 2 - 1)
    ref = O.forward_strands(ids, O.params_from_state_dict(sd, cfg))
    m = build(cfg, sd, torch.float32)
    out = m(input_ids=ids.to(DEV), output_hidden_states=True)
    lg = out.logits.cpu()
    assert lg.dtype == torch.float32 and lg.shape == (B, L, 8)
    scale = ref["logits"].abs().max()
    assert ((lg - ref["logits"]).abs().max() / scale).item() < 1e-4
    hid = out.hidden_states[-1].cpu()
    assert hid.shape == (B, L, 2 * D)
    assert len(out.hidden_states) == nl + 1                    # HF length; levels that were not kept raise
    with pytest.raises(IndexError, match="materialize_all_hidden_states"):
        out.hidden_states[0]
    assert ((hid - ref["hidden"]).abs().max() / ref["hidden"].abs().max()).item() < 1e-4
    p = L // 2 - 1
    assert torch.equal(lg[:, p, 3:7].argmax(-1), ref["logits"][:, p, 3:7].argmax(-1))
    # positions fast path == slicing the full output
    out_p = m(input_ids=ids.to(DEV), output_hidden_states=True, positions=[p, 0, L - 1])
    assert torch.equal(out_p.logits.cpu(), lg[:, [p, 0, L - 1]])
    assert torch.equal(out_p.hidden_states[-1].cpu(), hid[:, [p, 0, L - 1]])


@pytest.mark.parametrize("L", [128, 77])
def test_forward_bf16_vs_bf16_emulating_oracle(L):
    """bf16 path: compared with the oracle rounding to bf16 at the reference's tensor boundaries.
    Tolerance 3e-2 of the logit range (bf16 eps = 7.8e-3 through 4 layers), argmax exact where the
    top-2 margin exceeds that noise.  L = 77: ragged length (partial row blocks and tiles in every kernel)."""
    cfg = make_config("x", d_model=256, n_layer=4)
    sd = synthetic_state_dict(cfg, seed=3)
    ids = rand_ids(4, L, 9, mask=63)
    P = O.params_from_state_dict(sd, cfg, dtype=torch.bfloat16)
    ref = O.forward_strands(ids, P, rnd=O.round_bf16)
    ref32 = O.forward_strands(ids, O.params_from_state_dict(sd, cfg))
    m = build(cfg, sd, torch.bfloat16)
    out = m(input_ids=ids.to(DEV), output_hidden_states=True)
    lg = out.logits.cpu()
    assert out.hidden_states[-1].dtype == torch.bfloat16 and lg.dtype == torch.float32
    scale = ref["logits"].abs().max()
    err_emul = ((lg - ref["logits"]).abs().max() / scale).item()
    err_fp32 = ((lg - ref32["logits"]).abs().max() / scale).item()
    print(f"bf16 path: err vs bf16-emulating oracle {err_emul:.3e}, vs fp32 oracle {err_fp32:.3e}")
    assert err_emul < 3e-2
    top2 = ref["logits"][:, 63, 3:7].topk(2, dim=-1).values
    confident = (top2[:, 0] - top2[:, 1]) > 3e-2 * scale
    assert torch.equal(lg[:, 63, 3:7].argmax(-1)[confident], ref["logits"][:, 63, 3:7].argmax(-1)[confident])


def test_rc_equivariance_property_full_size():
    """size-independent property at the real l20 width/length: logits(rc x)[L-1-l, comp v] == logits(x)[l, v]
    and hidden(rc x) == flip_{seq,chan}(hidden(x)).  Exact in exact arithmetic; fp32 tolerance 1e-4."""
    cfg = make_config("l20", n_layer=2)
    sd = synthetic_state_dict(cfg, seed=1)
    ids = rand_ids(4, 512, 13)
    comp = torch.tensor(cfg.complement_list())
    rc = comp[ids.flip(-1)]
    m = build(cfg, sd, torch.float32)
    a = m(input_ids=ids.to(DEV), output_hidden_states=True)
    b = m(input_ids=rc.to(DEV), output_hidden_states=True)
    la, lb = a.logits.cpu(), b.logits.cpu()
    assert ((lb.flip(1)[:, :, comp] - la).abs().max() / la.abs().max()).item() < 1e-4
    ha, hb = a.hidden_states[-1].cpu(), b.hidden_states[-1].cpu()
    assert ((hb.flip(1, 2) - ha).abs().max() / ha.abs().max()).item() < 1e-4


def test_batch_chunking_and_empty_and_all_hidden():
    cfg = make_config("x", d_model=64, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=2)
    ids = rand_ids(7, 32, 21)
    m = build(cfg, sd, torch.float32, chunk_seqs=3)                  # 7 sequences -> chunks 3,3,1
    lg = m(input_ids=ids.to(DEV)).logits.cpu()
    m2 = build(cfg, sd, torch.float32, chunk_seqs=64)
    assert torch.equal(lg, m2(input_ids=ids.to(DEV)).logits.cpu())
    # the reference-order gate (each direction gated and rounded) is the same function in fp32 up to rounding
    m4 = build(cfg, sd, torch.float32, gate_each=1)
    lg4 = m4(input_ids=ids.to(DEV)).logits.cpu()
    assert ((lg4 - lg).abs().max() / lg.abs().max()).item() < 1e-5
    # ragged / empty
    assert m2(input_ids=ids[:0].to(DEV)).logits.shape == (0, 32, 8)
    # full hidden-state tuple vs literal-RCPS oracle
    cfg.materialize_all_hidden_states = True
    m3 = build(cfg, sd, torch.float32)
    out = m3(input_ids=ids.to(DEV), output_hidden_states=True)
    ref = O.forward_literal(ids, O.params_from_state_dict(sd, cfg), output_hidden_states=True)
    assert len(out.hidden_states) == cfg.n_layer + 1
    for got, want in zip(out.hidden_states, ref["all_hidden"]):
        assert ((got.cpu() - want).abs().max() / want.abs().max()).item() < 1e-4


def test_no_cpu_fallback():
    cfg = make_config("x", d_model=64, n_layer=1)
    sd = synthetic_state_dict(cfg, seed=2)
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(input_ids=rand_ids(1, 16, 0))


@pytest.mark.parametrize("dtype,D,L,B", [(torch.float32, 128, 45, 3), (torch.bfloat16, 384, 203, 5), (torch.bfloat16, 1024, 512, 3),
                                          (torch.float32, 64, 1, 2)])
def test_poisoned_workspace_gives_identical_results(dtype, D, L, B):
    """uninitialised-read screen: with every workspace byte pre-set to 0xFF (NaN) before each forward the outputs must be
    bit-identical to the normal run, at ragged lengths (partial row blocks / tiles in every kernel) and with several chunks."""
    cfg = make_config("x", d_model=D, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=11)
    ids = rand_ids(B, L, 3, mask=L // 2).to(DEV)
    ref = build(cfg, sd, dtype)(input_ids=ids, output_hidden_states=True)
    for opts in (dict(poison_workspace=1), dict(poison_workspace=1, chunk_seqs=2)):
        out = build(cfg, sd, dtype, **opts)(input_ids=ids, output_hidden_states=True)
        assert torch.isfinite(out.logits).all()
        assert torch.equal(out.logits, ref.logits) and torch.equal(out.hidden_states[-1], ref.hidden_states[-1])


def test_out_of_vocabulary_ids_raise():
    """the reference's nn.Embedding raises on an id outside the table; the engine must not alias it to another row silently.
    Detection is on the device (no host synchronisation inside forward): `check_status()` — what the host loops call where
    they read results back — raises, and so does a later forward once the flag has arrived; a clean batch afterwards is fine."""
    cfg = make_config("x", d_model=64, n_layer=1)
    m = build(cfg, synthetic_state_dict(cfg, seed=2), torch.float32)
    ids = rand_ids(2, 16, 0)
    good = m(input_ids=ids.to(DEV)).logits.clone()
    m.check_status()
    for bad in (8, -1, 1000):
        ids2 = ids.clone()
        ids2[1, 3] = bad
        m(input_ids=ids2.to(DEV), positions=[3])
        with pytest.raises(IndexError, match="outside"):
            m.check_status()
        m(input_ids=ids2.to(DEV))
        torch.cuda.synchronize()
        with pytest.raises(IndexError, match="outside"):         # the next forward reports the earlier one's flag
            m(input_ids=ids.to(DEV))
        assert torch.equal(m(input_ids=ids.to(DEV)).logits, good)
        m.check_status()
    # per-window positions (pcad_forward_at) outside the window: flagged instead of silently clamped
    eng = m._engine()
    for bad in (16, -1):
        pos = torch.tensor([3, bad], dtype=torch.int32, device=DEV)
        eng.forward(ids.to(DEV), positions=pos)
        with pytest.raises(IndexError, match="position"):
            eng.check_status()
    eng.forward(ids.to(DEV), positions=torch.tensor([3, 15], dtype=torch.int32, device=DEV))
    eng.check_status()
    # the host loop surfaces it too
    from plantcaduceus_amd import zero_shot
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    arr = ids.numpy().astype("int32").copy()
    arr[0, 5] = 9
    with pytest.raises(IndexError, match="outside"):
        zero_shot.extract_logits(m, arr, DEV, 7, CaduceusTokenizer())


def test_long_window_8192():
    """PlantCAD2-sized context (reference docs/zero-shot-eval.md:31,42: 8 192-bp windows, token_idx 4095): the
    kernels are length-generic (time-sequential scan, sliding-window conv).  fp32, one l20-wide layer pair, against
    the C oracle port."""
    from oracle.c_oracle import COracle
    cfg = make_config("x", d_model=384, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=17)
    L = 8192
    ids = rand_ids(2, L, 3, mask=4095)
    m = build(cfg, sd, torch.float32)
    out = m(input_ids=ids.to(DEV), output_hidden_states=True, positions=[4095, 0, L - 1])
    lg_ref, hid_ref = COracle(sd, cfg).forward(ids.numpy(), want_hidden=True)
    sel = [4095, 0, L - 1]
    lg = out.logits.cpu().numpy()
    assert np.abs(lg - lg_ref[:, sel]).max() / np.abs(lg_ref).max() < 1e-4
    hid = out.hidden_states[-1].cpu().numpy()
    assert np.abs(hid - hid_ref[:, sel]).max() / np.abs(hid_ref).max() < 1e-4
    assert (lg[:, 0, 3:7].argmax(-1) == lg_ref[:, 4095, 3:7].argmax(-1)).all()


@pytest.mark.parametrize("dtype,L,B,nl", [(torch.float32, 8192, 1, 1), (torch.bfloat16, 8192, 2, 2), (torch.bfloat16, 512, 6, 2)])
def test_plantcad2_large_geometry(dtype, L, B, nl):
    """PlantCAD2-Large's block (reference docs/PlantCAD2-overview.md:21: d_model 1536 => d_inner 3072, dt_rank 96) at full width
    on the fused path: conv + x_proj with 8 output fragments and one Wx slab (convx.hip NJ = 8), dt_proj K = 96 prefetched inside
    the scan, the 4-wave GEMMs at N = 6144 / 1536, at the 8 192-bp context (segmented scan for the single window) and at 512 bp.
    fp32 against the C oracle to north_star's 1e-4; bf16 against its bf16-emulating mode."""
    from oracle.c_oracle import COracle
    cfg = make_config("pc2-large", n_layer=nl)
    assert (cfg.d_model, cfg.d_inner, cfg.dt_rank) == (1536, 3072, 96)
    sd = synthetic_state_dict(cfg, seed=23, stress=True)
    p = L // 2 - 1
    ids = rand_ids(B, L, 3, mask=p)
    sel = [p, 0, L - 1]
    m = build(cfg, sd, dtype)
    eng = m._engine()
    eng.profile(1)
    out = m(input_ids=ids.to(DEV), output_hidden_states=True, positions=sel)
    torch.cuda.synchronize()
    st = {k: v[0] for k, v in eng.profile_read().items()}
    eng.profile(False)
    assert st["gemm_x_proj"] == 0 and st["conv1d_bidir"] == nl, st           # the fused conv + x_proj kernel ran, not the three-launch fallback
    lg, hid = out.logits.cpu().numpy(), out.hidden_states[-1].float().cpu().numpy()
    if dtype == torch.float32:
        lg_ref, hid_ref = COracle(sd, cfg, blas=True).forward(ids.numpy(), want_hidden=True)
        e_l = np.abs(lg - lg_ref[:, sel]).max() / np.abs(lg_ref).max()
        e_h = np.abs(hid - hid_ref[:, sel]).max() / np.abs(hid_ref).max()
        print(f"PlantCAD2-Large geometry fp32 L={L}: logits {e_l:.2e} hidden {e_h:.2e}")
        assert e_l < 1e-4 and e_h < 1e-4
        assert (lg[:, 0, 3:7].argmax(-1) == lg_ref[:, p, 3:7].argmax(-1)).all()
    else:
        lg_ref, hid_ref = COracle(sd, cfg, blas=True, dtype=torch.bfloat16, emulate_bf16=True).forward(ids.numpy(), want_hidden=True)
        e_l = np.abs(lg - lg_ref[:, sel]).max() / np.abs(lg_ref).max()
        e_h = np.abs(hid - hid_ref[:, sel]).max() / np.abs(hid_ref).max()
        print(f"PlantCAD2-Large geometry bf16 L={L}: logits {e_l:.2e} hidden {e_h:.2e} (vs the bf16-emulating oracle)")
        assert e_l < 3e-2 and e_h < 3e-2 and np.isfinite(lg).all()


@pytest.mark.parametrize("dtype,D,L,B", [(torch.float32, 256, 4096, 1), (torch.bfloat16, 384, 8192, 2), (torch.bfloat16, 768, 2048, 3),
                                          (torch.float32, 128, 2080, 2),
                                          # 512-bp windows in tiny batches (the reference's notebook runs B = 1): segments of 64 steps
                                          (torch.float32, 1024, 512, 1), (torch.bfloat16, 1024, 512, 3), (torch.float32, 384, 300, 2)])
def test_segmented_scan_equals_single_walk(dtype, D, L, B):
    """long windows, few strands: the scan cuts every strand into segments run by separate workgroups (zero-state pass, carry,
    real pass; csrc/kernels.hpp::scan_segments).  Same function as one workgroup walking the whole strand (`scan_segments` = 0):
    fp32 to 2e-5 of max (the carried decay is exp2(A * sum delta) instead of a product of per-step exps), bf16 to bf16 noise;
    L = 2080: last segment shorter than the others and a partial 32-step block.  Round 5: the same switch governs the K-split of
    the fused conv + x_proj kernel on launches of at most 64 row tiles (csrc/convx.hip::convx_ksplit: several blocks per row tile,
    partial x_proj sums added by convx_reduce_kernel), so the 512-bp cases compare split against unsplit walks of that kernel too
    (another fp32 summation order of x_dbl), with the partial-sum scratch poisoned."""
    cfg = make_config("x", d_model=D, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=31)
    ids = rand_ids(B, L, 3, mask=L // 2).to(DEV)
    pos = [L // 2, 0, L - 1, min(1031, L - 2)]
    a = build(cfg, sd, dtype)(input_ids=ids, output_hidden_states=True, positions=pos)
    b = build(cfg, sd, dtype, scan_segments=0)(input_ids=ids, output_hidden_states=True, positions=pos)
    la, lb = a.logits.float().cpu(), b.logits.float().cpu()
    ha, hb = a.hidden_states[-1].float().cpu(), b.hidden_states[-1].float().cpu()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(la).all()
    assert ((la - lb).abs().max() / lb.abs().max()).item() < tol
    assert ((ha - hb).abs().max() / hb.abs().max()).item() < tol
    if dtype == torch.float32:
        assert torch.equal(la[..., 3:7].argmax(-1), lb[..., 3:7].argmax(-1))
    # and poisoned: the segment scratch is written before it is read
    c = build(cfg, sd, dtype, poison_workspace=1)(input_ids=ids, output_hidden_states=True, positions=pos)
    assert torch.equal(c.logits, a.logits)


@pytest.mark.parametrize("dtype,D,L,B,opts", [
    (torch.float32, 256, 128, 5, {}), (torch.bfloat16, 256, 256, 40, {}), (torch.bfloat16, 1024, 512, 24, {}),
    (torch.float32, 384, 192, 7, {"gate_each": 1}), (torch.bfloat16, 512, 2048, 28, {"reference_order": 1}),
    (torch.float32, 256, 320, 40, {"f32_gemm_split": 1}), (torch.float32, 1024, 512, 20, {"f32_gemm_split": 1, "gate_each": 1}),
    (torch.bfloat16, 1536, 1024, 8, {})])
def test_pair_walk_equals_plain_walk(dtype, D, L, B, opts):
    """launches with few scan waves (csrc/kernels.hpp::scan_pair_wanted: more than the segmented form's bound, at most 3 584 per
    direction): both directions run in ONE launch and walk half a strand per launch - forward rows [0, L/2) + reverse rows
    [L/2, L), then the second halves from the kept states on top of the other direction's partial.  Same function as the plain
    two-launch walk (`scan_segments` = 0): fp32 to summation order (the y accumulation starts from the other direction's partial
    on half of the rows), bf16 to the rounding of one addend; the bf16 model's "gate_each" (each direction gated and rounded,
    then summed) is the same two rounded addends in either form: bit-identical (fp32: the gate's multiply is contracted into the
    add, so which addend is the stored one shows in the last bit).  Also against the oracle, with the hand-over scratch poisoned; dt_rank
    96 (D 1536) and the split-bf16 fp32 model (the second launch writes out_proj's [hi | lo] operand from BOTH directions)."""
    from plantcaduceus_amd.engine import load_library
    cfg = make_config("x", d_model=D, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=41)
    ids = rand_ids(B, L, 9, mask=L // 2)
    pos = [L // 2, 0, L - 1, L // 2 - 1]
    a = build(cfg, sd, dtype, **opts)(input_ids=ids.to(DEV), output_hidden_states=True)
    b = build(cfg, sd, dtype, scan_segments=0, **opts)(input_ids=ids.to(DEV), output_hidden_states=True)
    la, lb = a.logits.float().cpu(), b.logits.float().cpu()
    ha, hb = a.hidden_states[-1].float().cpu(), b.hidden_states[-1].float().cpu()
    assert torch.isfinite(la).all()
    if dtype == torch.bfloat16 and (opts.get("gate_each") or opts.get("reference_order")):
        assert torch.equal(la, lb) and torch.equal(ha, hb)          # two bf16-rounded addends, summed: the same in either form
    else:
        tol = 5e-6 if dtype == torch.float32 else 2e-2
        assert ((la - lb).abs().max() / lb.abs().max()).item() < tol
        assert ((ha - hb).abs().max() / hb.abs().max()).item() < tol
        assert not torch.equal(ha, hb) or dtype == torch.float32        # the pair form really ran (bf16: another rounding pattern)
    if dtype == torch.float32:
        ref = O.forward_strands(ids, O.params_from_state_dict(sd, cfg))
        assert ((la - ref["logits"]).abs().max() / ref["logits"].abs().max()).item() < 1e-4
        assert ((ha - ref["hidden"]).abs().max() / ref["hidden"].abs().max()).item() < 1e-4
    c = build(cfg, sd, dtype, poison_workspace=1, **opts)(input_ids=ids.to(DEV), output_hidden_states=True)
    assert torch.equal(c.logits.float().cpu(), la)
    # the positions path (last-layer shortcut): the LAST layer never takes the pair form, so that its shortened plain walks stay
    # bit-identical to the full layer on the evaluated rows
    d = build(cfg, sd, dtype, **opts)(input_ids=ids.to(DEV), positions=pos)
    assert torch.equal(d.logits.float().cpu(), la[:, pos])


def test_maximum_chunk_bit_identical_to_small_chunks():
    """maximum sizes: at the l32 width one chunk holds up to 1023 windows = 1 047 552 token-rows, whose x / xc / y tensors are just
    under 4 GiB each, so the kernels' unsigned 32-bit byte offsets run to the top of their range.  Rows are independent, so the
    result must be bit-identical to the same batch walked in 64-window chunks (and to a 512-window chunk: offsets past 2^31)."""
    cfg = make_config("l32", n_layer=1)
    sd = synthetic_state_dict(cfg, seed=8)
    ids = rand_ids(1023, 512, 77, mask=255)
    pos = [255, 0, 511]
    big = build(cfg, sd, torch.bfloat16, chunk_seqs=1023)        # (the default would cut 1 023 windows into 512 + 511: whole GEMM rounds)
    a = big(input_ids=ids[:512].to(DEV), output_hidden_states=True, positions=pos)       # one chunk of 512 windows
    b = big(input_ids=ids.to(DEV), output_hidden_states=True, positions=pos)             # ONE chunk of 1023 windows
    la, ha = a.logits.cpu(), a.hidden_states[-1].float().cpu()
    lb, hb = b.logits.cpu(), b.hidden_states[-1].float().cpu()
    del big, a, b
    torch.cuda.empty_cache()
    small = build(cfg, sd, torch.bfloat16, chunk_seqs=64)
    c = small(input_ids=ids.to(DEV), output_hidden_states=True, positions=pos)
    lc, hc = c.logits.cpu(), c.hidden_states[-1].float().cpu()
    assert torch.isfinite(lc).all()
    assert torch.equal(la, lc[:512]) and torch.equal(ha, hc[:512])
    assert torch.equal(lb, lc) and torch.equal(hb, hc)


def test_repeated_forward_is_bit_identical():
    """race screen for the counted-vmcnt / one-barrier pipelines (4-wave GEMM, conv+x_proj, scan): the same batch run
    six times at the l32 width must give the same bits every time."""
    cfg = make_config("l32", n_layer=2)
    m = build(cfg, synthetic_state_dict(cfg, seed=9), torch.bfloat16)
    ids = rand_ids(192, 512, 5, mask=255).to(DEV)
    ref = None
    for _ in range(6):
        out = m(input_ids=ids, output_hidden_states=True, positions=[255, 3, 508])
        cur = (out.logits.clone(), out.hidden_states[-1].clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,nl,B,L,pos", [(128, 2, 5, 64, [31]), (128, 3, 3, 45, [0]), (128, 2, 3, 45, [44, 7]),
                                          (256, 2, 4, 512, [255]), (256, 1, 2, 512, [3, 200, 511]), (256, 2, 3, 203, [101])])
def test_last_layer_shortcut_bit_identical(dtype, D, nl, B, L, pos):
    """SURVEY.md §7 step 6 (the reference's callers read one position: src/zero_shot_score.py:117, src/train_XGBoost.py:105):
    with a list of evaluated positions the last layer's scans stop at the furthest one and out_proj runs on the gathered rows
    only.  Must equal the full last layer (option off) and the slice of the all-positions forward, bit for bit."""
    cfg = make_config("x", d_model=D, n_layer=nl)
    sd = synthetic_state_dict(cfg, seed=11, stress=True)
    ids = rand_ids(B, L, 3, mask=pos[0]).to(DEV)
    m_on = build(cfg, sd, dtype)
    a = m_on(input_ids=ids, output_hidden_states=True, positions=pos)
    full = m_on(input_ids=ids, output_hidden_states=True)
    m_off = build(cfg, sd, dtype, last_layer_shortcut=0)
    b = m_off(input_ids=ids, output_hidden_states=True, positions=pos)
    assert torch.equal(a.logits, b.logits) and torch.equal(a.hidden_states[-1], b.hidden_states[-1])
    assert torch.equal(a.logits, full.logits[:, pos]) and torch.equal(a.hidden_states[-1], full.hidden_states[-1][:, pos])
    assert torch.isfinite(a.logits).all()
    # with every workspace byte poisoned first: the shortened walks / gathered rows read nothing they did not write
    m_p = build(cfg, sd, dtype, poison_workspace=1)
    c = m_p(input_ids=ids, output_hidden_states=True, positions=pos)
    assert torch.equal(a.logits, c.logits) and torch.equal(a.hidden_states[-1], c.hidden_states[-1])


@pytest.mark.parametrize("L,pos", [(128, None), (77, None), (128, [63]), (45, [44, 0])])
def test_reference_order_levels_bf16(L, pos):
    """pcad_set_option("reference_order", 1 / 2) (include/pcad.h) against the torch oracle rounding to bf16 where the reference's
    bf16 model stores a tensor, in the operation order each level claims:
      level 2 (strict)  BiMambaWrapper's own order - each direction gated and rounded, its own tied out_proj stored in bf16, then
                        `out + out_rev` rounded (oracle forward_strands(tie_fold=False));
      level 1           the same with the two tied out_proj calls folded by linearity (tie_fold=True);
      level 0           the shipped default.
    With the rounding points in the same places only fp32 summation order and exp / log implementations differ, so the level's
    own emulation must be the closest one to it (up to the one-ulp noise two restatements have), and all stay inside the 4-layer
    bf16 bar of test_forward_bf16_vs_bf16_emulating_oracle.  Covers ragged lengths (reference-order launches only), the
    last-layer shortcut's strict branch (shared positions) and hidden_states[-1]."""
    cfg = make_config("x", d_model=256, n_layer=4)
    sd = synthetic_state_dict(cfg, seed=3)
    mask = 63 if L > 63 else L - 1
    ids = rand_ids(4, L, 9, mask=mask)
    P = O.params_from_state_dict(sd, cfg, dtype=torch.bfloat16)
    ref_strict = O.forward_strands(ids, P, rnd=O.round_bf16, tie_fold=False)
    ref_fold = O.forward_strands(ids, P, rnd=O.round_bf16, tie_fold=True)
    scale = ref_strict["logits"].abs().max()
    sel = (lambda t: t[:, pos]) if pos is not None else (lambda t: t)
    err = {}
    for level in (0, 1, 2):
        m = build(cfg, sd, torch.bfloat16, reference_order=level)
        out = m(input_ids=ids.to(DEV), output_hidden_states=True, **({"positions": pos} if pos is not None else {}))
        lg, hid = out.logits.cpu(), out.hidden_states[-1].float().cpu()
        for name, ref in (("strict", ref_strict), ("fold", ref_fold)):
            err[level, name] = ((lg - sel(ref["logits"])).abs().max() / scale).item()
        hs = sel(ref_strict["hidden"]).float()
        assert ((hid - hs).abs().max() / hs.abs().max()).item() < 3e-2
    print(f"L={L} pos={pos}: max logit error / range by (level, emulation order): " + ", ".join(f"{k}: {v:.2e}" for k, v in err.items()))
    assert all(v < 3e-2 for v in err.values())
    ulp = 2.0 ** -8                                          # one bf16 ulp of the logit range
    assert err[2, "strict"] <= err[0, "strict"] + ulp        # the strict level is not further from the reference's order than the default
    assert err[1, "fold"] <= err[0, "fold"] + ulp


@pytest.mark.parametrize("pos", [None, [31, 5]])
def test_reference_order_strict_fp32(pos):
    """The strict level on the fp32 model: no rounding anywhere, so it must meet north_star's 1e-4 like the default, on logits,
    hidden_states[-1] and (all positions) every level of output_hidden_states; and chunking must not change a bit."""
    cfg = make_config("x", d_model=128, n_layer=3)
    sd = synthetic_state_dict(cfg, seed=11)
    ids = rand_ids(5, 64, 4, mask=31)
    ref = O.forward_literal(ids, O.params_from_state_dict(sd, cfg), output_hidden_states=True)
    m = build(cfg, sd, torch.float32, reference_order=2)
    kw = {"positions": pos} if pos is not None else {}
    out = m(input_ids=ids.to(DEV), output_hidden_states=True, **kw)
    lg, hid = out.logits.cpu(), out.hidden_states[-1].cpu()
    rl, rh = (ref["logits"][:, pos], ref["hidden"][:, pos]) if pos is not None else (ref["logits"], ref["hidden"])
    assert ((lg - rl).abs().max() / ref["logits"].abs().max()).item() < 1e-4
    assert ((hid - rh).abs().max() / ref["hidden"].abs().max()).item() < 1e-4
    m2 = build(cfg, sd, torch.float32, reference_order=2, chunk_seqs=2)
    out2 = m2(input_ids=ids.to(DEV), output_hidden_states=True, **kw)
    assert torch.equal(out2.logits.cpu(), lg) and torch.equal(out2.hidden_states[-1].cpu(), hid)
    if pos is None:
        cfg.materialize_all_hidden_states = True
        allh = build(cfg, sd, torch.float32, reference_order=2)(input_ids=ids.to(DEV), output_hidden_states=True).hidden_states
        assert len(allh) == len(ref["all_hidden"]) == cfg.n_layer + 1
        for i, (a, b) in enumerate(zip(allh, ref["all_hidden"])):
            assert ((a.cpu().float() - b).abs().max() / b.abs().max().clamp_min(1e-6)).item() < 1e-4, i


@pytest.mark.parametrize("D,nl,B,L", [(128, 2, 5, 64), (384, 2, 3, 203), (256, 3, 4, 128), (1024, 1, 2, 512)])
def test_f32_gemm_split_matches_oracle(D, nl, B, L):
    """"f32_gemm_split" 1 on the fp32 model (include/pcad.h): north_star's 1e-4 on logits and hidden states, exact arg-max, on
    shapes that take the 4-wave GEMM (whole 256-row tiles), the 8-wave one (ragged rows) and the 256 x 128 one (d_model 384);
    chunked and un-chunked runs bit-identical; the strict reference order composes with it; the bf16 model ignores the option."""
    cfg = make_config("x", d_model=D, n_layer=nl)
    sd = synthetic_state_dict(cfg, seed=D + nl)
    ids = rand_ids(B, L, 5, mask=L // 2 - 1)
    ref = O.forward_strands(ids, O.params_from_state_dict(sd, cfg))
    m = build(cfg, sd, torch.float32, f32_gemm_split=1)
    out = m(input_ids=ids.to(DEV), output_hidden_states=True)
    lg, hid = out.logits.cpu(), out.hidden_states[-1].cpu()
    e_l = ((lg - ref["logits"]).abs().max() / ref["logits"].abs().max()).item()
    e_h = ((hid - ref["hidden"]).abs().max() / ref["hidden"].abs().max()).item()
    print(f"f32_gemm_split D={D} L={L}: logits rel err {e_l:.2e}, hidden {e_h:.2e}")
    assert e_l < 1e-4 and e_h < 1e-4
    p = L // 2 - 1
    assert torch.equal(lg[:, p, 3:7].argmax(-1), ref["logits"][:, p, 3:7].argmax(-1))
    m2 = build(cfg, sd, torch.float32, f32_gemm_split=1, chunk_seqs=2)
    assert torch.equal(m2(input_ids=ids.to(DEV)).logits.cpu(), lg)
    m3 = build(cfg, sd, torch.float32, f32_gemm_split=1, reference_order=2)
    assert ((m3(input_ids=ids.to(DEV)).logits.cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max()).item() < 1e-4
    # unsegmented scans (what large batches run): the reverse scan writes out_proj's [hi | lo] operand itself when L % 8 == 0
    m4 = build(cfg, sd, torch.float32, f32_gemm_split=1, scan_segments=0)
    out4 = m4(input_ids=ids.to(DEV), output_hidden_states=True)
    assert ((out4.logits.cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max()).item() < 1e-4
    assert ((out4.hidden_states[-1].cpu() - ref["hidden"]).abs().max() / ref["hidden"].abs().max()).item() < 1e-4
    m5 = build(cfg, sd, torch.float32, f32_gemm_split=1, scan_segments=0, gate_each=1, poison_workspace=1)
    assert ((m5(input_ids=ids.to(DEV)).logits.cpu() - ref["logits"]).abs().max() / ref["logits"].abs().max()).item() < 1e-4
    mb = build(cfg, sd, torch.bfloat16, f32_gemm_split=1)
    mb0 = build(cfg, sd, torch.bfloat16)
    assert torch.equal(mb(input_ids=ids.to(DEV)).logits.cpu(), mb0(input_ids=ids.to(DEV)).logits.cpu())


def test_options_that_need_bind_time_copies_are_refused_afterwards():
    """"f32_gemm_split" 1 and "norm_fold" 1 on an fp32 model need weight copies that are packed when the weights are bound
    (include/pcad.h): set through config.engine_options they work; set on a bound engine the next forward fails loudly instead of
    silently running another form; turning them OFF afterwards is always possible."""
    cfg = make_config("x", d_model=256, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=4)
    ids = rand_ids(2, 64, 3).to(DEV)
    m = build(cfg, sd, torch.float32)
    ref = m(input_ids=ids).logits.cpu()
    eng = m._engine()
    for opt in ("f32_gemm_split", "norm_fold"):
        eng.set_option(opt, 1)
        with pytest.raises(RuntimeError, match="pcad_bind_weights"):
            m(input_ids=ids)
        eng.set_option(opt, 0)
        assert torch.equal(m(input_ids=ids).logits.cpu(), ref)
    m2 = build(cfg, sd, torch.float32, f32_gemm_split=1)
    a = m2(input_ids=ids).logits.cpu()
    m2._engine().set_option("f32_gemm_split", 0)                 # off again: the plain fp32 GEMMs, bit-identical to a plain engine
    assert torch.equal(m2(input_ids=ids).logits.cpu(), ref)
    assert ((a - ref).abs().max() / ref.abs().max()).item() < 1e-5
    # the bf16 model's DEFAULT form is the folded one: bound under "reference_order" 1 (no folded copies packed) and switched back to
    # the default afterwards, the forward is refused as well instead of silently running unfolded; an engine bound in
    # the default form may be switched to the reference order and back at will (its copies exist)
    ids256 = rand_ids(2, 128, 3).to(DEV)                         # whole 256-row tiles: the folded form engages
    mb = build(cfg, sd, torch.bfloat16, reference_order=1)
    r1 = mb(input_ids=ids256).logits.cpu()
    mb._engine().set_option("reference_order", 0)
    with pytest.raises(RuntimeError, match="pcad_bind_weights"):
        mb(input_ids=ids256)
    mb._engine().set_option("reference_order", 1)
    assert torch.equal(mb(input_ids=ids256).logits.cpu(), r1)
    md = build(cfg, sd, torch.bfloat16)
    d0 = md(input_ids=ids256).logits.cpu()
    md._engine().set_option("reference_order", 1)
    assert torch.equal(md(input_ids=ids256).logits.cpu(), r1)
    md._engine().set_option("reference_order", 0)
    assert torch.equal(md(input_ids=ids256).logits.cpu(), d0) and not torch.equal(d0, r1)


def test_workspace_limit_bounds_the_allocation_without_changing_results():
    """pcad_set_option("workspace_limit_mb"): the engine's workspace slab stays under the limit (smaller chunks) and the results are
    the unlimited run's, bit for bit; Engine.release_workspace() frees the slab and the next forward re-allocates it."""
    cfg = make_config("x", d_model=512, n_layer=2)
    sd = synthetic_state_dict(cfg, seed=6)
    ids = rand_ids(24, 256, 8, mask=100).to(DEV)
    for dtype in (torch.bfloat16, torch.float32):
        ref_m = build(cfg, sd, dtype)
        ref = ref_m(input_ids=ids, output_hidden_states=True)
        full = ref_m._engine()._ws.numel()
        m = build(cfg, sd, dtype, workspace_limit_mb=48)
        out = m(input_ids=ids, output_hidden_states=True)
        eng = m._engine()
        assert eng._ws.numel() <= (48 << 20) + 256 < full
        assert torch.equal(out.logits, ref.logits) and torch.equal(out.hidden_states[-1], ref.hidden_states[-1])
        eng.release_workspace()
        assert eng._ws is None
        assert torch.equal(m(input_ids=ids).logits, ref.logits) and eng._ws is not None
