#!/usr/bin/env python3
"""End-to-end rate of the in-silico-mutagenesis region sweep (BASELINE config 5's shape; reference pipelines/in-silico-mutagenesis ->
src/zero_shot_score.py -input-vcf) on this engine, host side included (run on the GPU box):

    synthetic FASTA (one chromosome) -> ism.sweep_region_to_vcf(model, fasta, chrom, start, stop, ...) -> VCF rows on disk

one masked forward per position (all four allele probabilities per forward; the reference's pipeline would issue three), windows
built from one FASTA fetch + one tokeniser pass per chunk, rows written per chunk.  Prints one JSON object.

    python tools/e2e_ism.py [--positions 16384] [--model l32]
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
    ap.add_argument("--positions", type=int, default=16384)
    ap.add_argument("--model", default="l32")
    args = ap.parse_args()
    import numpy as np
    import torch
    from plantcaduceus_amd import ism, zero_shot
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint

    td = tempfile.mkdtemp(prefix="pcad_ism_")
    snap = os.path.join(td, "snap")
    make_synthetic_checkpoint(snap, args.model, seed=1234, stress=False)
    n = args.positions
    rng = np.random.default_rng(1)
    chrom = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n + 2000)]).decode()
    fa = os.path.join(td, "g.fa")
    with open(fa, "w") as f:
        f.write(">chr1\n")
        for i in range(0, len(chrom), 60):
            f.write(chrom[i:i + 60] + "\n")
    model, tok = zero_shot.load_model_and_tokenizer(snap, "cuda:0")
    out = os.path.join(td, "ism.vcf")
    ism.sweep_region_to_vcf(model, fa, "chr1", 0, 2048, tok, "cuda:0", out, chunk=2048)          # warm-up (workspace, library)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = ism.sweep_region_to_vcf(model, fa, "chr1", 1000, 1000 + n, tok, "cuda:0", out, chunk=8192)
    dt = time.perf_counter() - t0
    assert rows == 3 * n
    print(json.dumps({"workload": "ism.sweep_region_to_vcf: %d positions of a synthetic chromosome, PlantCaduceus_%s bf16, chunk 8192, "
                                  "one masked forward per position, 3 VCF rows per position" % (n, args.model),
                      "positions": n, "rows_written": rows, "wall_s": round(dt, 3), "positions_per_s": round(n / dt, 1),
                      "rows_per_s": round(rows / dt, 1),
                      "note": "host included: indexed FASTA fetch, window ids by sliding-window view, worker-thread batches with pinned "
                              "async copies, scores and row formatting per chunk; the GPU-only rate of the same forwards is bench.py "
                              "--workload zeroshot (positions share index 255 here, so the last-layer shortcut applies)"}))


if __name__ == "__main__":
    main()
