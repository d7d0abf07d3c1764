#!/usr/bin/env python3
"""Offline census (run on the GPU box) of north_star's "bit-exact argmax token calls" for the bf16 path at full depth.

  part 1  N synthetic 512-bp windows through PlantCaduceus_l32 bf16 on the HIP engine vs the C oracle in fp32 and in its
          bf16-emulating mode with the reference's operation order: how many 4-way calls at the masked index differ, the oracle's
          top-2 probability margin of every differing window, and the histogram of all margins (a synthetic random-weight model
          is far less confident than a trained one — reference notebooks/examples.ipynb:296 records p = 0.97 for its example).
  part 2  the same model on checkpoint.harsh_state_dict, full depth, fp32 and bf16, against the oracle:
          2a  dt_proj x256 on the `stress=True` checkpoint (distinct fwd/rev parameters): ~12 % of the time steps take softplus's
              pass-through branch (delta + bias > 20); the stack stays well conditioned, north_star's 1e-4 is asserted;
          2b  in_proj / x_proj x4 and dt_proj x16: |x| >> 1 through the stack — a chaotic network in which two CPU restatements
              that differ only in summation order already disagree by percent; reported next to that noise floor, no bar.

    python tools/argmax_census.py [--n 128] [--n-emul 64] [--model l32] > profiles/r03_argmax_census.txt

Round 5: `--fixture tests/golden/census_<model>.npz` runs the census against the COMMITTED oracle runs (512 l32 / 256 l20 windows,
oracle/gen_census_golden.py) for the engine's three operation orders (default, "reference_order" 1 and 2) - seconds instead of the
oracle's minutes; tests/test_gpu_census.py asserts on the same numbers.
    python tools/argmax_census.py --model l32 --fixture tests/golden/census_l32.npz > profiles/r05_argmax_census_l32.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = 255


def softmax4(z):
    import numpy as np
    p = np.exp(z - z.max(1, keepdims=True))
    return p / p.sum(1, keepdims=True)


def first_layer_delta_fraction(sd, cfg, ids, thr):
    """delta_raw + bias > thr in the first mixer (torch oracle's operators on the host: embedding of both strands -> RMSNorm ->
    in_proj -> conv1d + SiLU -> x_proj -> dt_proj), both directions"""
    import torch
    from oracle import caduceus_oracle as O
    P = O.params_from_state_dict(sd, cfg)
    S = O.strands(torch.from_numpy(ids).long(), P.complement)
    h = P.emb[S]
    lp = P.layers[0]
    u, _ = O.rms_norm_fn(h, lp.norm_w, residual=None, eps=P.eps, prenorm=True, residual_in_fp32=P.residual_in_fp32)
    xz = torch.einsum("bld,ed->bel", u, lp.fwd.in_proj)
    E = lp.fwd.conv_w.shape[0]
    n = tot = 0
    for p, rev in ((lp.fwd, False), (lp.rev, True)):
        x = xz[:, :E].flip(dims=(2,)) if rev else xz[:, :E]
        xc = O.causal_conv1d_fn(x, p.conv_w, p.conv_b, activation="silu")
        R = p.dt_proj_w.shape[1]
        x_dbl = torch.einsum("bel,re->blr", xc, p.x_proj)
        delta = torch.einsum("blr,er->bel", x_dbl[..., :R], p.dt_proj_w) + p.dt_proj_b.float()[None, :, None]
        n += int((delta > thr).sum())
        tot += delta.numel()
    return n / tot


def hip_probs(cfg, sd, ids, batch=None, **engine_options):
    """softmax over a, c, g, t at the masked index for every window, bf16 model on the HIP engine with the given pcad_set_option
    values; `batch`: windows per forward (BASELINE config 2 runs l20 at 1024 - the census windows are then the head of a batch
    padded with further seeded windows, so that the engine sees that configuration's launch sizes)."""
    import numpy as np
    import torch
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    cfg.engine_options = dict(engine_options)
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    m = m.to(torch.bfloat16).to("cuda:0")
    n = len(ids)
    batch = batch or n
    outs = []
    for b0 in range(0, n, batch):
        blk = ids[b0:b0 + batch]
        if len(blk) < batch:
            pad = np.random.default_rng(99).integers(3, 7, size=(batch - len(blk), ids.shape[1])).astype(np.int32)
            pad[:, P] = 1
            blk = np.concatenate([blk, pad], 0)
        lg = m(input_ids=torch.from_numpy(blk).to("cuda:0"), positions=[P]).logits[:, 0].float().cpu().numpy()
        outs.append(lg)
    m.check_status()
    del m
    torch.cuda.empty_cache()
    return softmax4(np.concatenate(outs, 0)[:n, 3:7])


def margins(q):
    import numpy as np
    top2 = np.sort(q, 1)[:, -2:]
    return top2[:, 1] - top2[:, 0]


def compare(p, q):
    """-> dict(n, flips: indices where the 4-way calls differ, max_dp, flip_margins: q's top-2 margin at those windows,
    margins_all: q's top-2 margin of every window)"""
    import numpy as np
    n = min(len(p), len(q))
    p, q = p[:n], q[:n]
    flips = np.nonzero(p.argmax(1) != q.argmax(1))[0]
    m = margins(q)
    return dict(n=n, flips=flips, max_dp=float(np.abs(p - q).max()), flip_margins=m[flips], margins_all=m)


MODES = (("default", {}), ("reference_order=1", {"reference_order": 1}), ("reference_order=2", {"reference_order": 2}))


def census_from_fixture(model, fixture, batch=None, out=print):
    """The census against the committed oracle runs (tests/golden/census_<model>.npz, oracle/gen_census_golden.py): the HIP bf16
    engine in its three operation orders x {fp32 oracle, reference-order bf16 emulation}, next to the floors the CPU restatements
    have between THEMSELVES (eng vs ref: one reordering; ref_plainc vs ref: same rounding points, other fp32 summation order;
    ref vs f32: what bf16 storage itself does).  Returns the numbers tests/test_gpu_census.py asserts on."""
    import hashlib
    import numpy as np
    from oracle.gen_census_golden import census_windows
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    fx = np.load(fixture)
    n, seed = int(fx["meta"][0]), int(fx["meta"][1])
    ids = census_windows(n, seed)
    assert hashlib.sha1(ids.tobytes()).digest() == fx["ids_sha1"].tobytes(), "fixture was generated from other windows"
    cfg = make_config(model)
    assert (cfg.d_model, cfg.n_layer) == (int(fx["meta"][4]), int(fx["meta"][5]))
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    orc = {k: softmax4(fx["logits_" + k][:, 3:7]) for k in ("f32", "ref", "eng", "ref_plainc")}
    n = min(n, len(orc["f32"]), len(orc["ref"]))                 # a generator run that was cut short holds a common prefix
    ids = ids[:n]
    res = {"n": n, "n_eng": len(orc["eng"]), "n_plainc": len(orc["ref_plainc"]), "modes": {}}
    out(f"== arg-max census, PlantCaduceus_{model} (d_model={cfg.d_model}, n_layer={cfg.n_layer}) bf16, {n} synthetic 512-bp windows, mask index {P}, "
        f"checkpoint seed 1234" + (f", forwards of {batch} windows" if batch else ""))
    floors = {"eng_vs_ref": compare(orc["eng"], orc["ref"]), "plainc_vs_ref": compare(orc["ref_plainc"], orc["ref"]),
              "ref_vs_f32": compare(orc["ref"][:n], orc["f32"][:n]), "eng_vs_f32": compare(orc["eng"], orc["f32"])}
    res["floors"] = floors
    out("-- floors between CPU restatements (no GPU involved):")
    for k, what in (("eng_vs_ref", "bf16 emulation, tied out_proj folded, vs reference order (ONE reordering)"),
                    ("plainc_vs_ref", "reference order, plain-C GEMM vs host BLAS (same rounding points, other fp32 summation order)"),
                    ("ref_vs_f32", "reference-order bf16 emulation vs fp32"), ("eng_vs_f32", "folded bf16 emulation vs fp32")):
        c = floors[k]
        out(f"   {what}: {len(c['flips'])} of {c['n']} calls differ, max |dp| {c['max_dp']:.3e}"
            + (f", margins of the differing windows {[float(f'{m:.2e}') for m in c['flip_margins']]}" if len(c["flips"]) else ""))
    h, edges = np.histogram(margins(orc["f32"][:n]), bins=[0, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2, 1e-1, 1.0])
    out("   fp32 oracle's top-2 probability margin, windows per bin: " + ", ".join(f"[{edges[i]:g},{edges[i+1]:g}): {h[i]}" for i in range(len(h))))
    for name, opts in MODES:
        p = hip_probs(cfg, sd, ids, batch=batch, **opts)
        r = {"vs_ref": compare(p, orc["ref"][:n]), "vs_f32": compare(p, orc["f32"][:n]), "vs_eng": compare(p, orc["eng"])}
        r["vs_ref_on_eng_prefix"] = compare(p[:len(orc["eng"])], orc["ref"][:len(orc["eng"])])
        res["modes"][name] = r
        out(f"-- HIP engine, {name}:")
        for k, what in (("vs_ref", "reference-order bf16 emulation"), ("vs_f32", "fp32 oracle"), ("vs_eng", "folded bf16 emulation")):
            c = r[k]
            out(f"   vs {what}: {len(c['flips'])} of {c['n']} calls differ, max |dp| {c['max_dp']:.3e}"
                + (f"; differing windows {c['flips'].tolist()} with oracle margins {[float(f'{m:.2e}') for m in c['flip_margins']]}" if len(c["flips"]) else ""))
    d_sum = floors["plainc_vs_ref"]["max_dp"]
    out(f"-- resolved calls (oracle margin >= {d_sum:.2e} = what another fp32 summation order alone does to the reference-order emulation): "
        + "; ".join(f"{name}: {int((r['vs_ref']['flip_margins'] >= d_sum).sum())} vs reference order, {int((r['vs_f32']['flip_margins'] >= d_sum).sum())} vs fp32"
                    for name, r in res["modes"].items())
        + f"; CPU emulations among themselves {int((floors['eng_vs_ref']['flip_margins'] >= d_sum).sum())}, reference-order emulation vs fp32 "
          f"{int((floors['ref_vs_f32']['flip_margins'] >= d_sum).sum())}")
    return res


def census_on_snapshot(snapshot, tsv=None, n_seeded=2048, n_emul=256, out=print, use_gpu=None, oracle_block=64):
    """The census on a REAL snapshot directory (config.json + weights under the reference's key names) - tools/real_weights.sh step
    (ii): the windows of the reference's example table (examples/example_snp.tsv -> tests/golden/example_snp.tsv, the rows
    src/zero_shot_score.py:232 keeps) plus `n_seeded` seeded uniform windows, all masked at 255, through
      the fp32 C oracle (every window) and its reference-order bf16 emulation (the first `n_emul` of each set),
      the HIP engine as bf16 in its three operation orders, as fp32 and as fp32 + "f32_gemm_split" (when a ROCm device is present),
    reporting per set: differing 4-way calls, max |dp|, the oracle margins of the differing windows, the margin histogram - on a
    TRAINED model, whose margins are not the synthetic checkpoint's (profiles/r05_argmax_census_*.txt).  Returns the numbers."""
    import json
    import numpy as np
    import pandas as pd
    import torch
    from oracle.c_oracle import COracle
    from plantcaduceus_amd.checkpoint import load_state_dict
    from plantcaduceus_amd.configuration_caduceus import config_from_dict
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    cfg = config_from_dict(json.load(open(os.path.join(snapshot, "config.json"))))
    sd = load_state_dict(snapshot)
    tok = CaduceusTokenizer.from_pretrained(snapshot) if os.path.exists(os.path.join(snapshot, "vocab.json")) else CaduceusTokenizer()
    sets = {}
    tsv = tsv or os.path.join(ROOT, "tests", "golden", "example_snp.tsv")
    if os.path.exists(tsv):
        df = pd.read_csv(tsv, delimiter="\t")
        df = df[df["ref"].isin(list("ACGT")) & df["alt"].isin(list("ACGT"))]
        sets["example table (%d rows of %s)" % (len(df), os.path.basename(tsv))] = tok.encode_batch(df["sequences"].tolist(), mask_index=P).astype(np.int32)
    if n_seeded > 0:
        ids = np.random.default_rng(0).integers(3, 7, size=(n_seeded, 512)).astype(np.int32)
        ids[:, P] = tok.mask_token_id
        sets["%d seeded uniform windows (default_rng(0))" % n_seeded] = ids
    gpu = torch.cuda.is_available() if use_gpu is None else bool(use_gpu)
    out(f"== arg-max census on snapshot {snapshot}: d_model={cfg.d_model}, n_layer={cfg.n_layer}, mask index {P}; engine rows "
        + ("on cuda:0" if gpu else "SKIPPED (no ROCm device: oracle-vs-oracle rows only)"))

    def oracle(ids, **kw):
        co = COracle(sd, cfg, blas=True, **kw)
        return softmax4(np.concatenate([co.forward(ids[b0:b0 + oracle_block])[0][:, P, 3:7] for b0 in range(0, len(ids), oracle_block)], 0))

    def hip(ids, dtype, **opts):
        from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
        cfg.engine_options = dict(opts)
        m = CaduceusForMaskedLM(cfg)
        m.load_state_dict(sd, strict=False)
        m.tie_weights()
        m = m.to(dtype).to("cuda:0")
        lg = np.concatenate([m(input_ids=torch.from_numpy(ids[b0:b0 + 1024]).to("cuda:0"), positions=[P]).logits[:, 0].float().cpu().numpy()
                             for b0 in range(0, len(ids), 1024)], 0)
        m.check_status()
        del m
        torch.cuda.empty_cache()
        return softmax4(lg[:, 3:7])

    res = {}
    edges = [0, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2, 1e-1, 0.5, 1.0]
    for name, ids in sets.items():
        t0 = time.time()
        p32 = oracle(ids)
        ne = min(n_emul, len(ids))
        pref = oracle(ids[:ne], dtype=torch.bfloat16, emulate_bf16=True, ref_order=True) if ne else None
        out(f"-- {name}: fp32 oracle + reference-order bf16 emulation of the first {ne}: {time.time() - t0:.0f} s")
        h, _ = np.histogram(margins(p32), bins=edges)
        out("   fp32 oracle's top-2 probability margin, windows per bin: " + ", ".join(f"[{edges[i]:g},{edges[i+1]:g}): {h[i]}" for i in range(len(h)))
            + f"; median top probability {float(np.median(p32.max(1))):.3f}")
        r = {"n": len(ids), "margin_hist": h.tolist(), "rows": {}}
        rows = []
        if pref is not None:
            rows.append(("reference-order bf16 emulation vs fp32 oracle (no GPU)", compare(pref, p32[:ne])))
        if gpu:
            for mode, opts in MODES:
                pb = hip(ids, torch.bfloat16, **opts)
                rows.append((f"HIP bf16 {mode} vs fp32 oracle", compare(pb, p32)))
                if pref is not None:
                    rows.append((f"HIP bf16 {mode} vs reference-order emulation", compare(pb[:ne], pref)))
            rows.append(("HIP fp32 vs fp32 oracle", compare(hip(ids, torch.float32), p32)))
            rows.append(("HIP fp32 + f32_gemm_split vs fp32 oracle", compare(hip(ids, torch.float32, f32_gemm_split=1), p32)))
        for what, c in rows:
            out(f"   {what}: {len(c['flips'])} of {c['n']} calls differ, max |dp| {c['max_dp']:.3e}"
                + (f"; oracle margins of the differing windows {[float(f'{m:.2e}') for m in c['flip_margins']]}" if len(c["flips"]) else ""))
            r["rows"][what] = {"n": c["n"], "flips": len(c["flips"]), "max_dp": c["max_dp"]}
        res[name] = r
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshot", default=None, help="a real snapshot directory: census of the example table's windows + --n-seeded seeded windows "
                    "on its (trained) weights against the fp32 oracle (tools/real_weights.sh step ii) instead of parts 1 and 2")
    ap.add_argument("--tsv", default=None, help="with --snapshot: the table whose `sequences` are scored (default tests/golden/example_snp.tsv)")
    ap.add_argument("--n-seeded", type=int, default=2048)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--n-emul", type=int, default=64)
    ap.add_argument("--n-stress", type=int, default=8)
    ap.add_argument("--model", default="l32")
    ap.add_argument("--fixture", default=None, help="tests/golden/census_<model>.npz: census against the committed oracle runs "
                    "(three engine operation orders; no oracle run) instead of parts 1 and 2")
    ap.add_argument("--batch", type=int, default=0, help="with --fixture: windows per forward (0: all at once)")
    args = ap.parse_args()
    if args.snapshot:
        census_on_snapshot(args.snapshot, tsv=args.tsv, n_seeded=args.n_seeded, n_emul=args.n_emul if args.n_emul != 64 else 256)
        return
    if args.fixture:
        census_from_fixture(args.model, args.fixture, batch=args.batch or None)
        return
    import numpy as np
    import torch
    from oracle.c_oracle import COracle
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM

    def hip(cfg, sd, dtype, ids, **kw):
        m = CaduceusForMaskedLM(cfg)
        m.load_state_dict(sd, strict=False)
        m.tie_weights()
        m = m.to(dtype).to("cuda:0")
        out = m(input_ids=torch.from_numpy(ids).to("cuda:0"), positions=[P], **kw)
        lg = out.logits[:, 0].float().cpu().numpy()
        m.check_status()
        del m
        torch.cuda.empty_cache()
        return lg

    cfg = make_config(args.model)
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)             # the benchmark's checkpoint
    rng = np.random.default_rng(0)
    ids = rng.integers(3, 7, size=(args.n, 512)).astype(np.int32)
    ids[:, P] = 1
    print(f"== part 1: PlantCaduceus_{args.model} (d_model={cfg.d_model}, n_layer={cfg.n_layer}), {args.n} synthetic windows, mask index {P}, "
          f"synthetic checkpoint seed 1234")
    p_hip = softmax4(hip(cfg, sd, torch.bfloat16, ids)[:, 3:7])
    t0 = time.time()
    p_f32 = softmax4(COracle(sd, cfg, blas=True).forward(ids)[0][:, P, 3:7])
    print(f"fp32 C oracle: {time.time() - t0:.0f} s")
    ne = min(args.n_emul, args.n)
    t0 = time.time()
    p_ref = softmax4(COracle(sd, cfg, blas=True, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True).forward(ids[:ne])[0][:, P, 3:7])
    print(f"bf16-emulating C oracle, reference operation order, first {ne} windows: {time.time() - t0:.0f} s")
    for name, q, n in (("fp32 oracle", p_f32, args.n), ("bf16 emulation (reference order)", p_ref, ne)):
        a, b = p_hip[:n].argmax(1), q.argmax(1)
        flips = np.nonzero(a != b)[0]
        top2 = np.sort(q, 1)[:, -2:]
        margin = top2[:, 1] - top2[:, 0]
        print(f"-- HIP bf16 vs {name}: {n} windows, max |dp| {np.abs(p_hip[:n] - q).max():.3e}, calls that differ: {len(flips)}")
        for i in flips:
            print(f"   window {i}: oracle p = {np.round(q[i], 4).tolist()}  hip p = {np.round(p_hip[i], 4).tolist()}  oracle top-2 margin {margin[i]:.4f}")
        edges = [0, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 5e-2, 1e-1, 1.0]
        h, _ = np.histogram(margin, bins=edges)
        print("   top-2 probability margin of the oracle, windows per bin: " + ", ".join(f"[{edges[i]:g},{edges[i+1]:g}): {h[i]}" for i in range(len(h))))
        print(f"   smallest margin among agreeing windows {margin[a == b].min():.2e}; largest margin among differing windows "
              f"{(margin[flips].max() if len(flips) else 0.0):.2e}")
    # the two oracles against each other: the noise floor that any bf16 implementation sits in
    a, b = p_f32[:ne].argmax(1), p_ref.argmax(1)
    print(f"-- fp32 oracle vs bf16 emulation (no GPU involved): calls that differ {int((a != b).sum())} of {ne}, max |dp| {np.abs(p_f32[:ne] - p_ref).max():.3e}")

    from plantcaduceus_amd.checkpoint import harsh_state_dict
    ids4 = ids[:args.n_stress]
    for title, kw, bar in (("part 2a: harsh checkpoint (stress seed 21, distinct fwd/rev parameters; dt_proj x256)", {}, True),
                           ("part 2b: the same with the projections scaled too (in_proj, x_proj x4; dt_proj x16): a chaotic stack",
                            dict(proj_scale=4.0, dt_scale=16.0), False)):
        print(f"\n== {title}, {args.n_stress} windows, full depth")
        sd4 = harsh_state_dict(cfg, **kw)
        frac = first_layer_delta_fraction(sd4, cfg, ids4[:2], 20.0)
        print(f"first layer, both directions, 2 windows: fraction of (t, channel) elements with dt_proj(x_dbl) + bias > 20 "
              f"(softplus pass-through branch) = {frac:.4f}")
        assert frac > 0.01, "the scaled checkpoint does not reach the softplus pass-through branch visibly"
        lg32 = hip(cfg, sd4, torch.float32, ids4)
        lgbf = hip(cfg, sd4, torch.bfloat16, ids4)
        ref32 = COracle(sd4, cfg, blas=True).forward(ids4)[0][:, P]
        ref32b = COracle(sd4, cfg, blas=False).forward(ids4[:2])[0][:, P]       # same arithmetic, other summation order: the noise floor
        refbf = COracle(sd4, cfg, blas=True, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True).forward(ids4)[0][:, P]
        floor = np.abs(ref32[:2] - ref32b).max() / np.abs(ref32).max()
        e32 = np.abs(lg32 - ref32).max() / np.abs(ref32).max()
        print(f"fp32: HIP vs C oracle logits rel err {e32:.2e}   (two CPU restatements, BLAS vs plain-C summation order, differ by {floor:.2e}); "
              f"argmax equal on {int((lg32[:, 3:7].argmax(1) == ref32[:, 3:7].argmax(1)).sum())} of {len(ids4)}")
        pb, qb = softmax4(lgbf[:, 3:7]), softmax4(refbf[:, 3:7])
        print(f"bf16: HIP vs bf16 emulation (reference order) max |dp| {np.abs(pb - qb).max():.3e}; argmax equal on "
              f"{int((pb.argmax(1) == qb.argmax(1)).sum())} of {len(ids4)}; vs fp32 oracle max |dp| {np.abs(pb - softmax4(ref32[:, 3:7])).max():.3e}")
        if bar:
            assert e32 < 1e-4
    assert np.isfinite(lg32).all() and np.isfinite(lgbf).all()


if __name__ == "__main__":
    main()
