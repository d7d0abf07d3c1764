"""Child of tests/test_gpu_dist.py::test_rccl_multi_rank_equals_single_rank, started by torchrun with one rank per GPU
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, as the driver starts bench.py for N > 1).

Every rank runs ITS block of the 185 example windows through the HIP engine on its own device and joins the one
all_gather_into_tensor over RCCL (plantcaduceus_amd/sharding.py); each rank writes what it holds afterwards plus a report of
who it was, so the parent can check: N ranks, N distinct devices, backend nccl, identical full results everywhere.  The same for
the PlantCAD2 loops (plantcad2_eval.masked_probs / unmasked_probs: one gather per chunk of windows).

The engine runs with "scan_segments" 0: the small-launch forms (segmented scan, conv + x_proj K-split) are chosen from the batch
size of each pcad_forward call (include/pcad.h), and a rank of an N-rank run sees smaller batches than the one-rank run - with the
forms on, the two runs would differ by fp32 summation order and the parent's bit-equality assertion would be about that policy,
not about the sharding.
Not collected by pytest (leading underscore)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main(outdir):
    import numpy as np
    import pandas as pd
    import torch
    import torch.distributed as dist
    from plantcaduceus_amd import embeddings, sharding, zero_shot
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    dev = sharding.init_from_env("cuda:0")                       # cuda:LOCAL_RANK + the RCCL group, as the CLIs do
    try:
        cfg = make_config("x", d_model=128, n_layer=2)
        cfg.engine_options = {"scan_segments": 0}                # batch-size-independent launch forms: see the module docstring
        m = CaduceusForMaskedLM(cfg)
        m.load_state_dict(synthetic_state_dict(cfg, seed=5), strict=False)
        m.tie_weights()
        m = m.to(torch.bfloat16).to(dev)
        df = pd.read_csv(os.path.join(ROOT, "tests", "golden", "example_snp.tsv"), delimiter="\t")
        df = df[df["ref"].isin(list("ACGT")) & df["alt"].isin(list("ACGT"))]
        seqs = df["sequences"].tolist()                          # 185 rows: not divisible by 2, 4 or 8 (padded tail)
        tok = CaduceusTokenizer()
        p = zero_shot.extract_logits(m, seqs, dev, 255, tok, batch_size=64)
        e = embeddings.extract_embeddings(m, seqs, dev, 255, tok, batch_size=64)
        # PlantCAD2 evaluation loops, sharded the same way (reference src/zero-shot-eval.py:129-178); 3 windows per rank and gather
        from plantcaduceus_amd import plantcad2_eval as pe
        pe.GATHER_CHUNK = 3
        mp3 = pe.masked_probs(m, tok, seqs[:37], [255, 256, 257], dev, batch_size=8)
        un = pe.unmasked_probs(seqs[:21], tok, m, dev, batch_size=4)
        rank, ws = sharding.world()
        np.savez(os.path.join(outdir, f"r{rank}.npz"), p=p, e=e, mp3=mp3, un=un)
        with open(os.path.join(outdir, f"r{rank}.json"), "w") as f:
            json.dump({"rank": rank, "world": ws, "backend": dist.get_backend() if dist.is_initialized() else None,
                       "device": torch.cuda.current_device(), "bus": torch.cuda.get_device_properties(torch.cuda.current_device()).name,
                       "visible": torch.cuda.device_count()}, f)
    finally:
        sharding.shutdown()


if __name__ == "__main__":
    main(sys.argv[1])
