"""-m gpu: parity at the FULL depth of the BASELINE models (PlantCaduceus_l20 = 20 layers x 384, l32 = 32 layers x 1024),
HIP path (through the HF surface / C ABI) against the C oracle port on the same seeded inputs.

  fp32 model  north_star's tolerance as written: logits and hidden <= 1e-4 of the tensor max, exact 4-way argmax of the
              a/c/g/t call on every window, over all 512 positions.
  bf16 model  (what the reference's dtype policy selects on this GPU) against the oracle in its bf16-emulating mode in
              BOTH operation orders — the reference's (each direction's tied out_proj computed and rounded, then summed:
              `ref_order=True`) and the engine's (out_proj of the sum) — and against the fp32 oracle; since round 5 on the
              first 64 windows of the committed full-depth oracle runs (tests/golden/census_*.npz) instead of 16 / 32 windows
              run live, which is most of what the suite's time was (tests/test_gpu_census.py counts calls on all 512 / 256).
              Tolerance on the probabilities 1.2e-2 (32 layers of bf16 rounding-order noise); the argmax must be exact
              wherever the oracle's top-2 margin exceeds twice that tolerance, at least half of the windows must be that
              confident (non-vacuous), and the fraction of ALL windows whose call equals the fp32 oracle's is printed and
              bounded below.
"""
import numpy as np
import pytest
import torch

from oracle.c_oracle import COracle
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from oracle_cache import oracle_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
P = 255


def hip_model(cfg, sd, dtype, **engine_options):
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    cfg.engine_options = dict(engine_options)
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    return m.to(dtype).to(DEV)


def windows(n, seed):
    ids = np.random.default_rng(seed).integers(3, 7, size=(n, 512)).astype(np.int32)
    ids[:, P] = 1
    ids[0, 7] = 2            # an N -> [UNK]
    return ids


def softmax4(z):
    p = np.exp(z - z.max(1, keepdims=True))
    return p / p.sum(1, keepdims=True)


def census_oracle(size, n):
    """The first n windows of the committed full-depth oracle runs (tests/golden/census_<size>.npz, oracle/gen_census_golden.py:
    benchmark checkpoint seed 1234, windows default_rng(0), mask at 255) -> (ids [n, 512], {mode: probabilities [n, 4]}) for the
    modes f32 / ref (bf16 emulation, reference order) / eng (bf16 emulation, tied out_proj folded).  The same oracle, run once on
    the GPU box's host instead of inside every GPU test run (each l32 mode costs ~1.3 s per window on 128 threads);
    tests/test_oracle.py::test_census_fixture_is_not_stale ties the file to the oracle source."""
    import hashlib
    import os
    from oracle.gen_census_golden import census_windows
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"census_{size}.npz"))
    ids = census_windows(int(fx["meta"][0]), int(fx["meta"][1]))
    assert hashlib.sha1(ids.tobytes()).digest() == fx["ids_sha1"].tobytes()
    probs = {k: softmax4(fx["logits_" + k][:n, 3:7]) for k in ("f32", "ref", "eng")}
    assert all(len(v) == n for v in probs.values())
    return ids[:n], probs


@pytest.mark.parametrize("size,n", [("l20", 16), ("l32", 16)])
def test_full_depth_fp32(size, n):
    cfg = make_config(size)
    sd = synthetic_state_dict(cfg, seed=21, stress=True)      # distinct fwd/rev parameters, non-unit norm weights / D
    ids = windows(n, 5)
    lg_ref, hid_ref = oracle_forward((size, 21, True), sd, cfg, ids, want_hidden=True)
    m = hip_model(cfg, sd, torch.float32)
    out = m(input_ids=torch.from_numpy(ids).to(DEV), output_hidden_states=True)
    lg = out.logits.cpu().numpy()
    hid = out.hidden_states[-1].cpu().numpy()
    e_l = np.abs(lg - lg_ref).max() / np.abs(lg_ref).max()
    e_h = np.abs(hid - hid_ref).max() / np.abs(hid_ref).max()
    print(f"{size} fp32 full depth: logits rel err {e_l:.2e}, hidden rel err {e_h:.2e}")
    assert e_l < 1e-4 and e_h < 1e-4
    assert (lg[:, P, 3:7].argmax(-1) == lg_ref[:, P, 3:7].argmax(-1)).all()
    # and the call at EVERY position whose oracle margin is above fp32 summation-order noise
    z = np.sort(lg_ref[..., 3:7], -1)
    sure = (z[..., -1] - z[..., -2]) > 1e-4 * np.abs(lg_ref).max()
    assert sure.mean() > 0.99
    assert (lg[..., 3:7].argmax(-1)[sure] == lg_ref[..., 3:7].argmax(-1)[sure]).all()


@pytest.mark.parametrize("size,n", [("l20", 64), ("l32", 64)])
def test_full_depth_bf16_both_orders(size, n):
    cfg = make_config(size)
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)   # BASELINE configs 2/3's checkpoint
    ids, orc = census_oracle(size, n)                        # committed oracle runs (round 5; before: 16 / 32 windows run live)
    m = hip_model(cfg, sd, torch.bfloat16)
    lg = m(input_ids=torch.from_numpy(ids).to(DEV), positions=[P]).logits[:, 0].cpu().numpy()
    p_hip = softmax4(lg[:, 3:7])
    p_ref, p_eng, p_f32 = orc["ref"], orc["eng"], orc["f32"]   # reference order / tied out_proj folded / fp32
    d_ref, d_eng, d_f32 = (np.abs(p_hip - q).max() for q in (p_ref, p_eng, p_f32))
    d_orders = np.abs(p_ref - p_eng).max()
    agree = {k: float((p_hip.argmax(1) == q.argmax(1)).mean()) for k, q in (("ref", p_ref), ("eng", p_eng), ("f32", p_f32))}
    print(f"{size} bf16 x{n}: max|dp| vs ref-order emulation {d_ref:.2e}, vs engine-order {d_eng:.2e}, vs fp32 {d_f32:.2e}; "
          f"the two emulations differ by {d_orders:.2e}; argmax agreement {agree}")
    TOL = 1.2e-2        # max over 64 windows (1e-2 held on the 16 / 32 of earlier rounds; over 512 windows the census sees 1.06e-2)
    assert d_ref < TOL and d_eng < TOL
    # regression guard: the tighter bar of the earlier rounds, on the first 16 windows it was stated on
    d16 = max(np.abs(p_hip[:16] - p_ref[:16]).max(), np.abs(p_hip[:16] - p_eng[:16]).max())
    print(f"   first 16 windows: max|dp| vs the two emulations {d16:.2e}")
    assert d16 < 1e-2
    assert d_f32 < 2 * TOL
    for q in (p_ref, p_eng, p_f32):
        top2 = np.sort(q, 1)[:, -2:]
        conf = (top2[:, 1] - top2[:, 0]) > 2 * TOL
        assert conf.sum() >= n // 2, "vacuous argmax check: too few confident windows"
        assert (p_hip.argmax(1)[conf] == q.argmax(1)[conf]).all()
    assert agree["f32"] >= 0.85 and agree["ref"] >= 0.85


def test_full_depth_bf16_engine_options_against_reference_order():
    """l32, all 32 layers, bf16, the three operation orders the engine offers, each against the C port emulating bf16 storage in
    the REFERENCE's order (`ref_order=True`: each direction gated and rounded, each tied out_proj rounded, then summed):
      gate_each=1   the SiLU gate applied and rounded per direction as selective_scan_fn does - the configuration closest to the
                    reference's rounding points (only the tied out_proj fold remains); include/pcad.h says this option restores the
                    reference's order, so it must be NO FURTHER from that emulation than the default order is (up to the noise
                    floor two bf16 restatements have between themselves);
      default       (here: norm_fold=0) gate applied once to the bi-directional sum, the reference's add + RMSNorm launch;
      norm_fold=1   add + RMSNorm folded into out_proj's epilogue / in_proj (the engine's shipped default).
    All three must stay inside the same 1e-2 bar on the probabilities and make the same confident calls."""
    cfg = make_config("l32")
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    n = 64
    ids, orc = census_oracle("l32", n)                       # the committed oracle runs test_full_depth_bf16_both_orders uses
    tids = torch.from_numpy(ids).to(DEV)
    p_ref, p_eng = orc["ref"], orc["eng"]
    floor = np.abs(p_ref - p_eng).max()                      # what reordering alone does to a bf16 restatement
    d = {}
    for name, opts in (("default", dict(norm_fold=0)), ("gate_each", dict(gate_each=1, norm_fold=0)), ("norm_fold", dict(norm_fold=1)),
                       ("gate_each+norm_fold", dict(gate_each=1, norm_fold=1))):
        lg = hip_model(cfg, sd, torch.bfloat16, **opts)(input_ids=tids, positions=[P]).logits[:, 0].cpu().numpy()
        p = softmax4(lg[:, 3:7])
        d[name] = float(np.abs(p - p_ref).max())
        top2 = np.sort(p_ref, 1)[:, -2:]
        conf = (top2[:, 1] - top2[:, 0]) > 2e-2
        assert conf.sum() >= n // 2
        assert (p.argmax(1)[conf] == p_ref.argmax(1)[conf]).all(), name
    print(f"l32 bf16 x{n}: max|dp| vs the reference-order emulation {d}; the two emulations differ by {floor:.2e}")
    assert all(v < 1.2e-2 for v in d.values())
    assert d["gate_each"] <= d["default"] + floor            # the reference-order option is not further from the reference's order
    assert d["norm_fold"] <= 2 * max(d["default"], floor) + 1e-3


@pytest.mark.parametrize("size,n", [("l20", 16), ("l32", 16)])
def test_full_depth_fp32_split_gemm(size, n):
    """north_star's 1e-4 (and exact arg-max) on the fp32 model at full depth with pcad_set_option("f32_gemm_split", 1): in_proj and
    out_proj as split-bf16 GEMMs on the bf16 matrix pipes (three bf16 products per fp32 product, fp32 accumulation) - the
    parity configuration at about twice the fp32-MFMA model's speed.  Same checkpoint, windows and oracle run as
    test_full_depth_fp32 (memoised)."""
    cfg = make_config(size)
    sd = synthetic_state_dict(cfg, seed=21, stress=True)
    ids = windows(n, 5)
    lg_ref, hid_ref = oracle_forward((size, 21, True), sd, cfg, ids, want_hidden=True)
    m = hip_model(cfg, sd, torch.float32, f32_gemm_split=1)
    out = m(input_ids=torch.from_numpy(ids).to(DEV), output_hidden_states=True)
    lg, hid = out.logits.cpu().numpy(), out.hidden_states[-1].cpu().numpy()
    e_l = np.abs(lg - lg_ref).max() / np.abs(lg_ref).max()
    e_h = np.abs(hid - hid_ref).max() / np.abs(hid_ref).max()
    print(f"{size} fp32 full depth, f32_gemm_split=1: logits rel err {e_l:.2e}, hidden rel err {e_h:.2e}")
    assert e_l < 1e-4 and e_h < 1e-4
    assert (lg[:, P, 3:7].argmax(-1) == lg_ref[:, P, 3:7].argmax(-1)).all()
    z = np.sort(lg_ref[..., 3:7], -1)
    sure = (z[..., -1] - z[..., -2]) > 1e-4 * np.abs(lg_ref).max()
    assert (lg[..., 3:7].argmax(-1)[sure] == lg_ref[..., 3:7].argmax(-1)[sure]).all()
    # the positions fast path (last-layer shortcut: the gathered rows go through the SAME split-bf16 product, same operand values and
    # K order as the full-size out_proj) is bit-identical to slicing the full output (round 5 ran them through the fp32 MFMA GEMM)
    lgp = m(input_ids=torch.from_numpy(ids).to(DEV), positions=[P]).logits[:, 0].cpu().numpy()
    np.testing.assert_array_equal(lgp, lg[:, P])


def test_full_depth_fp32_norm_fold():
    """north_star's 1e-4 on the fp32 model, all 32 layers, with the folded add + norm forced on (norm_fold=1; the fp32 model's
    default is the reference-order launch, which test_full_depth_fp32 runs: accumulating the GEMM onto the residual value costs
    fp32 precision - measured 2.2e-5 of max on the hidden states against 1.3e-6 - and only the bf16 model folds by default)."""
    cfg = make_config("l32")
    sd = synthetic_state_dict(cfg, seed=21, stress=True)
    ids16 = windows(16, 5)                                   # test_full_depth_fp32[l32-16]'s oracle run, first 8 windows of it
    lg_ref, hid_ref = (a[:8] for a in oracle_forward(("l32", 21, True), sd, cfg, ids16, want_hidden=True))
    ids = ids16[:8]
    out = hip_model(cfg, sd, torch.float32, norm_fold=1)(input_ids=torch.from_numpy(ids).to(DEV), output_hidden_states=True)
    lg, hid = out.logits.cpu().numpy(), out.hidden_states[-1].cpu().numpy()
    e_l = np.abs(lg - lg_ref).max() / np.abs(lg_ref).max()
    e_h = np.abs(hid - hid_ref).max() / np.abs(hid_ref).max()
    print(f"l32 fp32 full depth, norm_fold=1: logits rel err {e_l:.2e}, hidden rel err {e_h:.2e}")
    assert e_l < 1e-4 and e_h < 1e-4
    assert (lg[:, P, 3:7].argmax(-1) == lg_ref[:, P, 3:7].argmax(-1)).all()


def test_full_depth_harsh_checkpoint():
    """l32 at full depth on `harsh_state_dict` (dt_proj x256 on the stress checkpoint): ~12 % of the time steps in softplus's
    pass-through branch, delta + bias > 20 (asserted > 1 % on the first layer) — the regime the benign benchmark checkpoint never
    reaches.  fp32 to north_star's 1e-4 (two CPU restatements agree to 6e-8 on this checkpoint, so the bar is meaningful; with
    the projections scaled x4 as well the network amplifies summation-order noise to 4e-2 and no such bar exists: see
    checkpoint.harsh_state_dict and profiles/r03_argmax_census.txt); bf16 against the reference-order emulation."""
    import importlib.util
    import os
    from plantcaduceus_amd.checkpoint import harsh_state_dict
    spec = importlib.util.spec_from_file_location("argmax_census", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "argmax_census.py"))
    census = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(census)
    cfg = make_config("l32")
    sd = harsh_state_dict(cfg)
    ids = windows(8, 9)
    assert census.first_layer_delta_fraction(sd, cfg, ids[:2], 20.0) > 0.01
    tids = torch.from_numpy(ids).to(DEV)
    lg32 = hip_model(cfg, sd, torch.float32)(input_ids=tids, positions=[P]).logits[:, 0].cpu().numpy()
    ref32 = COracle(sd, cfg, blas=True).forward(ids)[0][:, P]
    e = np.abs(lg32 - ref32).max() / np.abs(ref32).max()
    lgbf = hip_model(cfg, sd, torch.bfloat16)(input_ids=tids, positions=[P]).logits[:, 0].cpu().numpy()
    refbf = COracle(sd, cfg, blas=True, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True).forward(ids)[0][:, P]
    pb, qb, pf = softmax4(lgbf[:, 3:7]), softmax4(refbf[:, 3:7]), softmax4(ref32[:, 3:7])
    d_he, d_hf, d_ef = np.abs(pb - qb).max(), np.abs(pb - pf).max(), np.abs(qb - pf).max()
    print(f"harsh checkpoint l32: fp32 logits rel err {e:.2e}; bf16 max|dp| HIP vs reference-order emulation {d_he:.2e}, HIP vs fp32 "
          f"{d_hf:.2e}, emulation vs fp32 {d_ef:.2e}")
    assert np.isfinite(lg32).all() and np.isfinite(lgbf).all()
    assert e < 1e-4
    assert (lg32[:, 3:7].argmax(-1) == ref32[:, 3:7].argmax(-1)).all()
    # bf16 with a tenth of the time steps on the pass-through branch: rounding to bf16 moves the probabilities by percents here
    # (the emulation's own distance from fp32, d_ef, is the yardstick: no bf16 implementation can be closer to another than
    # that scale).  The HIP path must sit inside that noise — not further from fp32 or from the emulation than twice the
    # emulation is from fp32 — and make the same call wherever the fp32 margin exceeds it.
    noise = 2 * d_ef + 1e-2
    assert d_hf < noise and d_he < noise
    top2 = np.sort(pf, 1)[:, -2:]
    sure = (top2[:, 1] - top2[:, 0]) > 2 * noise
    assert (pb.argmax(1)[sure] == pf.argmax(1)[sure]).all()
