#!/usr/bin/env python3
"""Arg-max census fixtures — TEST INFRASTRUCTURE ONLY (generator of tests/golden/census_<model>.npz).

north_star asks for "bit-exact argmax token calls"; for a bf16 path that can only be judged against the noise two bf16
restatements of the SAME network have between themselves.  This script runs the C oracle (oracle/c) at FULL depth on N seeded
synthetic 512-bp windows in the modes the parity tests compare the HIP engine with, and stores the 8 logits at the masked index:

    f32      fp32 arithmetic, fp32 weights                                    (north_star's "reference CPU path")
    ref      bf16 storage emulated, the REFERENCE's operation order            (COracle(ref_order=True): each direction gated and
             rounded, each tied out_proj rounded, then summed - BiMambaWrapper "add"; rms_norm_fn as its own rounding point)
    eng      bf16 storage emulated, tied out_proj folded                       (COracle(ref_order=False): = the engine's
             "reference_order" level 1; differs from `ref` by ONE reordering: the floor any reordering makes)
    ref_plainc  as `ref` with the plain-C GEMM loops instead of the host BLAS  (first --n-plainc windows: same rounding points,
             other fp32 summation order - the floor that two BLAS libraries have between them)

Full depth costs ~1.3 s per l32 window and mode on 128 host threads, so this runs ONCE (on the GPU box's host cores:
`gpurun -- python oracle/gen_census_golden.py --model l32 --n 512 --out gpurun_out/census_l32.npz`) and the result is committed;
tests/test_gpu_census.py compares the HIP engine with it in seconds, tests/test_oracle.py re-derives a small sample of it with
the oracle on whatever host runs the CPU suite (fixture not stale).  Inputs: windows = default_rng(seed).integers(3, 7) with
[MASK] (id 1) at index 255 - the generator of tools/argmax_census.py and bench.py; checkpoint = synthetic_state_dict(cfg, 1234,
stress=False), the benchmark's.  Results are written after every chunk (atomic rename), so a run cut short keeps its prefix.
"""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = 255


def census_windows(n, seed=0, L=512):
    import numpy as np
    ids = np.random.default_rng(seed).integers(3, 7, size=(n, L)).astype(np.int32)
    ids[:, P] = 1
    return ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="l32")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--n-eng", type=int, default=-1, help="windows for the `eng` mode (-1: all)")
    ap.add_argument("--n-plainc", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import numpy as np
    import torch
    from oracle.c_oracle import COracle
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict

    cfg = make_config(args.model)
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    ids = census_windows(args.n, args.seed)
    n_eng = args.n if args.n_eng < 0 else min(args.n, args.n_eng)
    modes = [("f32", args.n, dict(blas=True)),
             ("ref", args.n, dict(blas=True, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True)),
             ("eng", n_eng, dict(blas=True, dtype=torch.bfloat16, emulate_bf16=True, ref_order=False)),
             ("ref_plainc", min(args.n, args.n_plainc), dict(blas=False, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True))]
    src = open(os.path.join(ROOT, "oracle", "c", "pcad_oracle.c"), "rb").read()
    out = {"ids_sha1": np.frombuffer(hashlib.sha1(ids.tobytes()).digest(), dtype=np.uint8),
           "oracle_sha1": np.frombuffer(hashlib.sha1(src).digest(), dtype=np.uint8),
           "meta": np.array([args.n, args.seed, P, 1234, cfg.d_model, cfg.n_layer], dtype=np.int64)}
    secs = {}

    def save():
        tmp = args.out + ".tmp.npz"
        np.savez(tmp, **out, **{"seconds_" + k: np.float64(v) for k, v in secs.items()})
        os.replace(tmp, args.out)

    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    # round-robin over the modes chunk by chunk, so that a run cut short still holds every mode on a common prefix
    oracles = {name: COracle(sd, cfg, **kw) for name, _, kw in modes}
    print(f"{args.model}: {args.n} windows, {oracles['f32'].threads} host threads", flush=True)
    for name, n, _ in modes:
        out["logits_" + name] = np.zeros((0, 8), dtype=np.float32)
        secs[name] = 0.0
    for c0 in range(0, args.n, args.chunk):
        for name, n, _ in modes:
            c1 = min(c0 + args.chunk, n)
            if c1 <= c0:
                continue
            t0 = time.time()
            lg = oracles[name].forward(ids[c0:c1])[0][:, P, :]
            secs[name] += time.time() - t0
            out["logits_" + name] = np.concatenate([out["logits_" + name], lg.astype(np.float32)], axis=0)
        save()
        print(f"  windows [0, {min(c0 + args.chunk, args.n)}): " + ", ".join(f"{k} {v:.0f} s" for k, v in secs.items()), flush=True)
    print("done: " + ", ".join(f"{k}: {out['logits_' + k].shape[0]} windows" for k, _, _ in modes))


if __name__ == "__main__":
    main()
