#!/usr/bin/env python3
"""End-to-end wall clock of the reference's command line on this engine (run on the GPU box):

    python -m plantcaduceus_amd.zero_shot -input-table <5 000 rows> -output out.tsv -model <snapshot> -device cuda:0

The reference's only published measurement is exactly this (5 000 SNPs, 512-bp windows, end to end: README.md:331-332 and the
table at :375-384 — PlantCaduceus_l32 47 s on one H100, other hardware, context only).  A synthetic l32 snapshot
(config.json + model.safetensors under the reference's key names, seed 1234) and a 5 000-row table of random windows are
written to a scratch directory; `zero_shot.main` runs in THIS process with its stages timed by wrapping the functions it calls
(nothing mocked).  Prints one JSON object; `profiles/r03_e2e_5000.json` is a committed copy.

    python tools/e2e_5000.py [--rows 5000] [--model l32] [--batchSize N]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5000)
    ap.add_argument("--model", default="l32")
    ap.add_argument("--batchSize", type=int, default=None)
    ap.add_argument("--keep", default=None, help="directory to build the snapshot / table in (default: a temporary one)")
    args = ap.parse_args()

    import numpy as np
    import pandas as pd
    import torch
    from plantcaduceus_amd import zero_shot
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint

    td = args.keep or tempfile.mkdtemp(prefix="pcad_e2e_")
    os.makedirs(td, exist_ok=True)
    snap = os.path.join(td, "snap")
    t0 = time.perf_counter()
    if not os.path.exists(os.path.join(snap, "config.json")):
        make_synthetic_checkpoint(snap, args.model, seed=1234, stress=False)
    t_make = time.perf_counter() - t0
    rng = np.random.default_rng(0)
    n = args.rows
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [bytes(letters[rng.integers(0, 4, size=512)]).decode() for _ in range(n)]
    ref = [s[255] for s in seqs]
    alt = ["ACGT"[("ACGT".index(r) + 1 + int(k)) % 4] for r, k in zip(ref, rng.integers(0, 3, size=n))]
    inp, out = os.path.join(td, "in.tsv"), os.path.join(td, "out.tsv")
    pd.DataFrame({"chr": "1", "pos": np.arange(1, n + 1), "ref": ref, "alt": alt, "sequences": seqs}).to_csv(inp, sep="\t", index=False)

    # stage timers: wrap the functions zero_shot.main calls
    stages = {}

    def timed(name, fn, sync=False):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            if sync:
                torch.cuda.synchronize()
            stages[name] = stages.get(name, 0.0) + time.perf_counter() - t
            return r
        return w

    zero_shot.load_model_and_tokenizer = timed("load_model_s", zero_shot.load_model_and_tokenizer, sync=True)
    zero_shot.extract_logits = timed("extract_logits_s", zero_shot.extract_logits)
    zero_shot.tokenize_masked = timed("tokenise_s (worker thread, overlapped)", zero_shot.tokenize_masked)
    zero_shot.zero_shot_score = timed("scores_s", zero_shot.zero_shot_score)
    _read = pd.read_csv
    pd.read_csv = timed("read_table_s", _read)
    _to_csv = pd.DataFrame.to_csv
    pd.DataFrame.to_csv = timed("write_table_s", _to_csv)

    argv = ["-input-table", inp, "-output", out, "-model", snap, "-device", "cuda:0"]
    if args.batchSize:
        argv += ["-batchSize", str(args.batchSize)]
    torch.cuda.init()
    t0 = time.perf_counter()
    zero_shot.main(argv)
    total = time.perf_counter() - t0
    pd.read_csv, pd.DataFrame.to_csv = _read, _to_csv
    res = pd.read_csv(out, sep="\t")
    assert len(res) == n and np.isfinite(res["zeroShotScore"]).all()
    print(json.dumps({
        "command": "python -m plantcaduceus_amd.zero_shot " + " ".join(argv[:1] + ["<%d rows>" % n] + argv[2:3] + ["out.tsv", "-model", "<synthetic %s snapshot>" % args.model, "-device", "cuda:0"]),
        "rows": n, "model": args.model, "total_wall_s": round(total, 3), "rows_per_s_end_to_end": round(n / total, 1),
        "stages_s": {k: round(v, 3) for k, v in stages.items()},
        "not_in_total": {"make_synthetic_snapshot_s": round(t_make, 2)},
        "note": "stages are wall clock of the wrapped calls inside zero_shot.main; tokenisation runs on a worker thread while the "
                "GPU works (its time is inside extract_logits_s, not added to it); load_model_s includes reading the snapshot, "
                "packing the weights on the device and the first import of the HIP library; reference README.md:375-384 reports "
                "47 s for the same command with PlantCaduceus_l32 on one H100 (other hardware, real weights; context only)"}))


if __name__ == "__main__":
    main()
