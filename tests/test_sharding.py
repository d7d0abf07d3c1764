"""CPU: the N>1 path (block sharding + one all-gather) with world_size 2 over gloo, asserting the gathered
rows are identical in order and content to the single-rank run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plantcaduceus_amd import sharding


def test_shard_bounds_cover_and_pad():
    for n in (0, 1, 7, 8, 9, 185, 100000):
        for ws in (1, 2, 4, 8):
            spans = [sharding.shard_bounds(n, r, ws) for r in range(ws)]
            per = spans[0][2]
            assert per * ws >= n and (n == 0 or per * ws - n < ws)
            cover = [i for a, b, _ in spans for i in range(a, b)] if n < 1000 else None
            if cover is not None:
                assert cover == list(range(n))
    x = torch.arange(6).reshape(3, 2)
    assert sharding.pad_rows(x, 5).shape == (5, 2) and torch.equal(sharding.pad_rows(x, 5)[4], x[2])
    assert sharding.pad_rows(x[:0], 2).shape == (2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from oracle import caduceus_oracle as O
        from plantcaduceus_amd import embeddings, zero_shot
        from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
        from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
        torch.set_num_threads(2)
        cfg = make_config("x", d_model=32, n_layer=1)
        model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=1), cfg))
        model.config = cfg
        rng = np.random.default_rng(0)
        seqs = ["".join(rng.choice(list("ACGT"), size=24)) for _ in range(n)]
        tok = CaduceusTokenizer()
        p = zero_shot.extract_logits(model, seqs, "cpu", 11, tok, batch_size=2)
        e = embeddings.extract_embeddings(model, seqs, "cpu", 11, tok, batch_size=2)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), p=p, e=e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 0])
def test_two_rank_gather_equals_single_rank(tmp_path, n):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert r0["p"].shape == (n, 4) and r0["e"].shape[0] == n
    np.testing.assert_array_equal(r0["p"], r1["p"])           # every rank holds the full result
    np.testing.assert_array_equal(r0["e"], r1["e"])
    if n:
        # single-rank run in this process
        from oracle import caduceus_oracle as O
        from plantcaduceus_amd import embeddings, zero_shot
        from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
        from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
        cfg = make_config("x", d_model=32, n_layer=1)
        model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=1), cfg))
        rng = np.random.default_rng(0)
        seqs = ["".join(rng.choice(list("ACGT"), size=24)) for _ in range(n)]
        tok = CaduceusTokenizer()
        np.testing.assert_allclose(zero_shot.extract_logits(model, seqs, "cpu", 11, tok, batch_size=2), r0["p"], rtol=1e-6)
        np.testing.assert_allclose(embeddings.extract_embeddings(model, seqs, "cpu", 11, tok, batch_size=2), r0["e"], rtol=1e-6, atol=1e-7)
