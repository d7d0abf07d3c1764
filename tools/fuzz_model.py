"""developer tool (run on the GPU box): randomized model-level parity — random geometry (d_model, layers, dt_rank, batch,
ragged sequence lengths, positions), fp32 against the torch oracle (1e-4 of max, exact argmax where the margin exceeds noise)
and bf16 against the bf16-emulating oracle (3e-2 of the logit range).      python tools/fuzz_model.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import caduceus_oracle as O
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM

from oracle.c_oracle import usable_cpus
if usable_cpus():                       # container CPU quota below the visible CPUs: a larger torch team is throttled as a group
    torch.set_num_threads(usable_cpus())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
# "fold" as a third argument: geometries on which the norm-folded layer form engages (d_model % 256 == 0, 2 B L % 256 == 0), the
# option forced on for both dtypes - tile counts from 1 to a few hundred, fewer tiles than CUs, 1..4 layers
FOLD = len(sys.argv) > 3 and sys.argv[3] == "fold"
# "opts" as a third argument (round 5): random engine options on top of the random geometry - "reference_order" 0 / 1 / 2,
# "f32_gemm_split" (fp32 cases), "scan_segments", "chunk_seqs", "workspace_limit_mb" - every combination must stay inside the same bars
OPTS = len(sys.argv) > 3 and sys.argv[3] == "opts"
bad = 0
t0 = time.time()
for case in range(n):
    D = int(rng.choice([64, 128, 192, 256, 384, 512, 768]))
    nl = int(rng.integers(1, 4))
    B = int(rng.integers(1, 6))
    L = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 257, 300]))
    if FOLD:
        D = int(rng.choice([256, 512, 768, 1024, 320, 384, 448]))       # incl. widths padded to the next multiple of 256
        nl = int(rng.integers(1, 5))
        L = int(rng.choice([32, 64, 128, 256, 384, 512, 640]))
        B = int(rng.integers(1, 9)) * (128 // np.gcd(128, L))            # 2 B L a multiple of 256
    bf16 = bool(rng.integers(0, 2))
    ssm = dict(d_state=16, d_conv=4, expand=2, dt_rank=int(rng.choice([D // 16, max(1, D // 16 - 1), min(64, D // 16 + 3), 80, 96, 70])), bias=False, conv_bias=True)
    cfg = make_config("x", d_model=D, n_layer=nl, ssm_cfg=ssm)
    sd = synthetic_state_dict(cfg, seed=int(rng.integers(0, 1 << 30)), stress=True)
    ids = torch.from_numpy(rng.integers(0, 8, size=(B, L)))          # every token id incl. PAD / MASK / UNK / the pad row
    dt = torch.bfloat16 if bf16 else torch.float32
    if FOLD:
        cfg.engine_options = {"norm_fold": 1}
    split = False
    if OPTS:
        if rng.integers(0, 2):                                            # half of the cases on widths / lengths where every fast path engages
            D = int(rng.choice([256, 512, 768, 1024, 384]))
            L = int(rng.choice([64, 128, 256, 512]))
            ssm["dt_rank"] = int(rng.choice([D // 16, 48, 64, 96]))
            cfg = make_config("x", d_model=D, n_layer=nl, ssm_cfg=ssm)
            sd = synthetic_state_dict(cfg, seed=int(rng.integers(0, 1 << 30)), stress=True)
            ids = torch.from_numpy(rng.integers(0, 8, size=(B, L)))
        eo = {"reference_order": int(rng.integers(0, 3)), "scan_segments": int(rng.integers(0, 2))}
        if rng.integers(0, 2):
            eo["chunk_seqs"] = int(rng.integers(1, B + 1))
        if rng.integers(0, 3) == 0:
            eo["workspace_limit_mb"] = int(rng.choice([1, 8, 64]))
        split = (not bf16) and bool(rng.integers(0, 2))
        if split:
            eo["f32_gemm_split"] = 1
        cfg.engine_options = eo
    m = CaduceusForMaskedLM(cfg); m.load_state_dict(sd, strict=False); m.tie_weights(); m = m.to(dt).to("cuda:0")
    out = m(input_ids=ids.to("cuda:0"), output_hidden_states=True)
    lg, hid = out.logits.cpu(), out.hidden_states[-1].float().cpu()
    P = O.params_from_state_dict(sd, cfg, dtype=dt)
    ref = O.forward_strands(ids, P, rnd=O.round_bf16 if bf16 else O._ident, tie_fold=bf16)
    scale = ref["logits"].abs().max().clamp_min(1e-20)
    e_l = ((lg - ref["logits"]).abs().max() / scale).item()
    e_h = ((hid - ref["hidden"]).abs().max() / ref["hidden"].abs().max().clamp_min(1e-20)).item()
    tol = 3e-2 if bf16 else 1e-4
    ok = e_l < tol and e_h < tol and bool(torch.isfinite(lg).all())
    pos = sorted(set(int(p) for p in rng.integers(0, L, size=min(L, 3))))
    outp = m(input_ids=ids.to("cuda:0"), output_hidden_states=True, positions=pos)
    # the last-layer shortcut is bit-identical to the full layer in every configuration (round 6: also with f32_gemm_split, whose
    # gathered rows now go through the same split-bf16 product; the pair walks never take the last layer)
    ok = ok and torch.equal(outp.logits.cpu(), lg[:, pos]) and torch.equal(outp.hidden_states[-1].float().cpu(), hid[:, pos])
    if not ok:
        bad += 1
    print(f"case {case:3d} D={D:3d} nl={nl} R={cfg.dt_rank:2d} B={B} L={L:3d} {'bf16' if bf16 else 'fp32'}  logits {e_l:.2e} hidden {e_h:.2e}  {'ok' if ok else 'FAIL'}"
          + (f"  {cfg.engine_options}" if OPTS else ""))
    del m
print(f"{n} random cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
