#!/usr/bin/env python3
"""End-to-end rate of the embedding-extraction path (BASELINE config 4; reference src/train_XGBoost.py:96-114,175-190) for one rank's
share of a 100 000-window `-chunk_size` chunk on one GPU (run on the GPU box): window strings -> embeddings.extract_embeddings ->
fp32 [N, d_model] averaged embeddings -> .npz cache with the reference's key.  Prints one JSON object.

    python tools/e2e_embed.py [--windows 12500] [--model l32]
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
    ap.add_argument("--windows", type=int, default=12500)
    ap.add_argument("--model", default="l32")
    args = ap.parse_args()
    import numpy as np
    import torch
    from plantcaduceus_amd import embeddings, zero_shot
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    td = tempfile.mkdtemp(prefix="pcad_emb_")
    snap = os.path.join(td, "snap")
    cfg, _ = make_synthetic_checkpoint(snap, args.model, seed=1234, stress=False)
    rng = np.random.default_rng(2)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [bytes(letters[rng.integers(0, 4, size=512)]).decode() for _ in range(args.windows)]
    model, tok = zero_shot.load_model_and_tokenizer(snap, "cuda:0")
    embeddings.extract_embeddings(model, seqs[:1024], "cuda:0", 255, tok)                # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = embeddings.extract_embeddings(model, seqs, "cuda:0", 255, tok)
    t1 = time.perf_counter()
    embeddings.save_embedding_cache(os.path.join(td, "cache.npz"), train=emb)
    t2 = time.perf_counter()
    assert emb.shape == (args.windows, cfg.d_model) and np.isfinite(emb).all()
    print(json.dumps({"workload": "embeddings.extract_embeddings: %d synthetic 512-bp windows (one rank's share of a 100 000-window chunk), "
                                  "PlantCaduceus_%s bf16, hidden_states[-1] at index 255, (fwd + channel-reversed rc)/2 -> fp32 [N, %d]"
                                  % (args.windows, args.model, cfg.d_model),
                      "windows": args.windows, "extract_s": round(t1 - t0, 3), "windows_per_s": round(args.windows / (t1 - t0), 1),
                      "save_npz_compressed_s": round(t2 - t1, 3),
                      "note": "host included (tokenisation on a worker thread, pinned async copies, one D2H of the [N, d_model] result); "
                              "np.savez_compressed with the reference's cache key is timed separately"}))


if __name__ == "__main__":
    main()
