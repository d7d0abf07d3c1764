#!/usr/bin/env python3
"""One command for the day real weights exist (there are none offline: every parity statement of this repo is against an
oracle whose RCPS wiring is restated from memory of the upstream modeling code - SURVEY.md §8(c), "parity unpinned").

    tools/real_weights.sh <snapshot_dir>            # e.g. a download of kuleshov-group/PlantCaduceus_l20
    python tools/real_weights.py <snapshot_dir> [--steps audit,known,census,e2e] [--n-seeded 2048] [--out DIR]

Steps (each prints PASS / FAIL / SKIP and the run ends non-zero if any FAILed):
  audit   checkpoint.audit_snapshot: every key of config.json is one this implementation consumes (unknown keys FAIL - PretrainedConfig
          would swallow them silently), the configuration is one the engine implements, every tensor of the module tree
          (reference notebooks/examples.ipynb:61-100) is present with its shape, tied pairs are tied, complement_map buffers agree
          with the config.
  known   PLANTCAD_L20_DIR=<dir> pytest tests/test_known_answer.py: the reference's ONE recorded answer (notebooks/examples.ipynb:296,
          l20, sequence :142, mask :258-284) through the oracle's literal and 2B-strand forms and, on a GPU box, the HIP engine.
          Only meaningful for the PlantCaduceus_l20 snapshot (d_model 384, 20 layers): other geometries SKIP.
  census  tools/argmax_census.py --snapshot: the example table's 185 windows + 2 048 seeded windows on the TRAINED weights - differing
          4-way calls and the oracle's margin histogram for the engine's three bf16 operation orders, fp32 and fp32 + f32_gemm_split
          against the fp32 oracle (engine rows need a ROCm device; without one the oracle-vs-oracle rows are still produced).
  e2e     the reference's command line (src/zero_shot_score.py -input-table ... -model <dir>) on the example table, end to end, on
          the GPU; rows / s and the head of the scores (SKIP without a ROCm device).
Without a directory (or with one that does not exist) every step SKIPs and the exit code is 0, so the script can sit in a
round-end checklist.  tests/test_real_weights.py drives it on a synthetic snapshot (CPU) so that the script itself is exercised.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = ("audit", "known", "census", "e2e")


def _gpu() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def step_audit(snap, args, log):
    from plantcaduceus_amd.checkpoint import audit_snapshot
    rep = audit_snapshot(snap, strict=False)
    log(json.dumps({k: rep[k] for k in ("geometry", "tensors", "config_keys")}, indent=1))
    if rep["problems"]:
        for p in rep["problems"]:
            log("PROBLEM: " + p)
        return "FAIL", "%d problem(s): %s" % (len(rep["problems"]), rep["problems"][0][:160])
    return "PASS", "config keys consumed %d, bookkeeping %d, generic %d; %d tensors" % (
        len(rep["config_keys"]["consumed"]), len(rep["config_keys"]["bookkeeping"]), len(rep["config_keys"]["hf_generic"]), rep["tensors"]["in_file"])


def step_known(snap, args, log):
    raw = json.load(open(os.path.join(snap, "config.json")))
    if (raw.get("d_model"), raw.get("n_layer")) != (384, 20):
        return "SKIP", "the recorded answer is PlantCaduceus_l20's (d_model 384, 20 layers); this snapshot is %s / %s" % (raw.get("d_model"), raw.get("n_layer"))
    env = dict(os.environ, PLANTCAD_L20_DIR=snap)
    marker = "gpu or not gpu" if _gpu() else "not gpu"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_known_answer.py"), "-q", "-m", marker, "-rs"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    log(r.stdout[-3000:] + r.stderr[-1500:])
    tail = [ln for ln in r.stdout.strip().splitlines() if ln.strip()][-1] if r.stdout.strip() else ""
    return ("PASS" if r.returncode == 0 else "FAIL"), tail


def step_census(snap, args, log):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import argmax_census
    res = argmax_census.census_on_snapshot(snap, tsv=args.tsv, n_seeded=args.n_seeded, n_emul=args.n_emul, out=log)
    if args.out:
        with open(os.path.join(args.out, "census.json"), "w") as f:
            json.dump(res, f, indent=1)
    worst = max((row["flips"] / max(1, row["n"]) for s in res.values() for k, row in s["rows"].items() if k.startswith("HIP fp32")), default=None)
    if worst is not None and worst > 0:
        return "FAIL", "the fp32 engine differs from the fp32 oracle in %.2f %% of the calls" % (100 * worst)
    return "PASS", "; ".join("%s: %s" % (k.split(":")[0][:24], {kk[:28]: vv["flips"] for kk, vv in s["rows"].items()}) for k, s in res.items())[:400]


def step_e2e(snap, args, log):
    if not _gpu():
        return "SKIP", "no ROCm device"
    import numpy as np
    import pandas as pd
    from plantcaduceus_amd import zero_shot
    tsv = args.tsv or os.path.join(ROOT, "tests", "golden", "example_snp.tsv")
    outp = os.path.join(args.out or "/tmp", "real_weights_scores.tsv")
    t0 = time.perf_counter()
    zero_shot.main(["-input-table", tsv, "-output", outp, "-model", snap, "-device", "cuda:0"])
    dt = time.perf_counter() - t0
    df = pd.read_csv(outp, sep="\t")
    ok = len(df) > 0 and np.isfinite(df["zeroShotScore"]).all()
    log(df.head(8).to_string())
    return ("PASS" if ok else "FAIL"), "%d rows in %.1f s end to end (model load included); scores finite: %s; written to %s" % (len(df), dt, ok, outp)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("snapshot", nargs="?", default=os.environ.get("PLANTCAD_SNAPSHOT", ""))
    ap.add_argument("--steps", default=",".join(STEPS))
    ap.add_argument("--tsv", default=None, help="table with a `sequences` column (default tests/golden/example_snp.tsv)")
    ap.add_argument("--n-seeded", type=int, default=2048)
    ap.add_argument("--n-emul", type=int, default=256)
    ap.add_argument("--out", default=None, help="directory for census.json / the score table / the full log")
    args = ap.parse_args(argv)
    steps = [s for s in args.steps.split(",") if s]
    bad = [s for s in steps if s not in STEPS]
    if bad:
        ap.error("unknown step(s) %s (of %s)" % (bad, list(STEPS)))
    if args.out:
        os.makedirs(args.out, exist_ok=True)
    lines = []

    def log(msg):
        print(msg, flush=True)
        lines.append(str(msg))

    snap = args.snapshot
    have = bool(snap) and os.path.isdir(snap) and os.path.exists(os.path.join(snap, "config.json"))
    if not have:
        log("real_weights: no snapshot directory given (or %r has no config.json): every step SKIPs.  Usage: tools/real_weights.sh <dir>" % snap)
    results = {}
    for s in steps:
        if not have:
            results[s] = ("SKIP", "no snapshot")
            continue
        log("==== %s" % s)
        t0 = time.time()
        try:
            results[s] = globals()["step_" + s](snap, args, log)
        except Exception as ex:            # a step that cannot run is a failure of that step, not of the script
            import traceback
            log(traceback.format_exc())
            results[s] = ("FAIL", repr(ex)[:300])
        log("---- %s: %s (%s) [%.0f s]" % (s, results[s][0], results[s][1], time.time() - t0))
    log("==== summary: " + ", ".join("%s %s" % (s, results[s][0]) for s in steps))
    if args.out:
        with open(os.path.join(args.out, "real_weights.log"), "w") as f:
            f.write("\n".join(lines) + "\n")
    return 1 if any(v[0] == "FAIL" for v in results.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
