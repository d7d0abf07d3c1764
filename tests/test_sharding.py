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
        torch.set_num_threads(1 if ws > 2 else 2)
        cfg = make_config("x", d_model=32, n_layer=1)
        model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=1), cfg))
        model.config = cfg
        rng = np.random.default_rng(0)
        seqs = ["".join(rng.choice(list("ACGT"), size=24)) for _ in range(n)]
        tok = CaduceusTokenizer()
        seen = []                                     # rows this rank tokenises (must be its own block only)
        enc = tok.encode_batch
        tok.encode_batch = lambda ss, mask_index=None: (seen.append(len(ss)), enc(ss, mask_index=mask_index))[1]
        p = zero_shot.extract_logits(model, seqs, "cpu", 11, tok, batch_size=16)
        e = embeddings.extract_embeddings(model, seqs, "cpu", 11, tok, batch_size=16)
        a, b, _ = sharding.shard_bounds(n, rank, ws)
        # batch by batch (the next batch is tokenised on a worker thread while the current one runs), own block only, twice
        assert sum(seen) == 2 * (b - a) and all(0 < k <= 16 for k in seen), (seen, a, b)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), p=p, e=e)
    finally:
        dist.destroy_process_group()


def _decide_worker(rank, ws, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        # each rank "sees" something else (a cache file visible on some ranks only): everyone must follow rank 0
        got = [sharding.rank0_decides(rank == 0), sharding.rank0_decides(rank != 0), sharding.rank0_decides(rank % 2 == 1)]
        with open(os.path.join(outdir, f"d{rank}.txt"), "w") as f:
            f.write(repr(got))
    finally:
        dist.destroy_process_group()


def test_rank0_decides_is_uniform_across_ranks(tmp_path):
    """a yes / no that selects between a branch with collectives and one without (cache file exists?) is rank 0's
    value on every rank, whatever each rank sees; outside a process group it is the caller's own value."""
    assert sharding.rank0_decides(True) is True and sharding.rank0_decides(False) is False
    mp.spawn(_decide_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert open(tmp_path / f"d{r}.txt").read() == repr([True, False, False])


@pytest.mark.parametrize("ws,n", [(2, 5), (2, 0), (4, 185), (4, 1), (8, 185), (8, 1), (8, 0)])
def test_multi_rank_gather_equals_single_rank(tmp_path, ws, n):
    """world 2 / 4 / 8 over gloo; N = 185 (the example table: not divisible by 4 or 8, tail padded), N = 1 (fewer rows than
    ranks: most ranks run only the dummy row) and N = 0."""
    port = _free_port()
    mp.spawn(_worker, args=(ws, port, n, str(tmp_path)), nprocs=ws, join=True)
    rs = [np.load(tmp_path / f"r{r}.npz") for r in range(ws)]
    r0 = rs[0]
    assert r0["p"].shape == (n, 4) and r0["e"].shape[0] == n
    for r in rs[1:]:
        np.testing.assert_array_equal(r0["p"], r["p"])           # every rank holds the full result
        np.testing.assert_array_equal(r0["e"], r["e"])
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
        np.testing.assert_allclose(zero_shot.extract_logits(model, seqs, "cpu", 11, tok, batch_size=16), r0["p"], rtol=1e-6)
        np.testing.assert_allclose(embeddings.extract_embeddings(model, seqs, "cpu", 11, tok, batch_size=16), r0["e"], rtol=1e-6, atol=1e-7)


def _pc2_setup(n, L=24):
    from oracle import caduceus_oracle as O
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=2), cfg))
    model.config = cfg
    rng = np.random.default_rng(3)
    seqs = ["".join(rng.choice(list("ACGT"), size=L)) for _ in range(n)]
    return model, CaduceusTokenizer(), seqs


def _pc2_worker(rank, ws, port, n, chunk, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from plantcaduceus_amd import plantcad2_eval as pe
        torch.set_num_threads(1)
        pe.GATHER_CHUNK = chunk                          # several gathers per call (the shipped chunk is 4 096 windows per rank)
        model, tok, seqs = _pc2_setup(n)
        seen = []
        enc = tok.encode_batch
        tok.encode_batch = lambda ss, mask_index=None: (seen.append(len(ss)), enc(ss, mask_index=mask_index))[1]
        m1 = pe.masked_probs(model, tok, seqs, 11, "cpu", batch_size=4)
        m3 = pe.masked_probs(model, tok, seqs, [13, 11, 12], "cpu", batch_size=5)
        un = pe.unmasked_probs(seqs, tok, model, "cpu", batch_size=4)
        a, b, _ = sharding.shard_bounds(n, rank, ws)
        assert sum(seen) == 3 * (b - a), (seen, a, b)    # each rank tokenises its own block only, once per call
        np.savez(os.path.join(outdir, f"q{rank}.npz"), m1=m1, m3=m3, un=un)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ws,n,chunk", [(2, 11, 3), (4, 11, 2), (4, 3, 4096), (2, 0, 4)])
def test_plantcad2_loops_sharded_equal_single_process(tmp_path, ws, n, chunk):
    """plantcad2_eval.masked_probs / unmasked_probs (reference src/zero-shot-eval.py:129-178, single-device loops) under a process
    group: every rank evaluates its block, ONE all-gather per chunk of windows (chunk forced small here: several gathers, ragged
    last chunk, ranks with no rows at all), every rank returns all rows - bit-equal to the single-process run, in its order."""
    mp.spawn(_pc2_worker, args=(ws, _free_port(), n, chunk, str(tmp_path)), nprocs=ws, join=True)
    from plantcaduceus_amd import plantcad2_eval as pe
    model, tok, seqs = _pc2_setup(n)
    ref = dict(m1=pe.masked_probs(model, tok, seqs, 11, "cpu", batch_size=4), m3=pe.masked_probs(model, tok, seqs, [13, 11, 12], "cpu", batch_size=5),
               un=pe.unmasked_probs(seqs, tok, model, "cpu", batch_size=4))
    assert ref["m1"].shape == (n, 4) and ref["m3"].shape == (3 * n, 4) and ref["un"].shape == (n, 24 if n else 0, 4)
    for r in range(ws):
        got = np.load(tmp_path / f"q{r}.npz")
        for k in ("m1", "m3", "un"):
            if n:
                np.testing.assert_allclose(got[k], ref[k], rtol=1e-6, atol=1e-7)     # batch composition differs -> BLAS summation order may
            assert got[k].shape == ref[k].shape
            np.testing.assert_array_equal(got[k], np.load(tmp_path / "q0.npz")[k])    # all ranks hold identical arrays


@pytest.mark.gpu
def test_rccl_all_gather_branch_single_rank():
    """the GPU branch of sharding.all_gather_rows (dist.all_gather_into_tensor over RCCL) executed once: world 1 on cuda:0.
    (`world() == 1` short-circuits in all_gather_rows, so sharding._all_gather — what world > 1 runs — is called directly.)"""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        local = torch.arange(12, dtype=torch.float32, device=dev).reshape(3, 4)
        out = torch.empty_like(local)
        dist.all_gather_into_tensor(out, local.contiguous())
        torch.cuda.synchronize()
        assert torch.equal(out, local)
        # and through the product's collective function (all_gather_rows skips it when world() == 1)
        emb = torch.randn(5, 1024, device=dev)
        assert torch.equal(sharding._all_gather(local, 1), local)
        assert torch.equal(sharding._all_gather(emb, 1), emb)
        assert sharding.world() == (0, 1) and torch.equal(sharding.all_gather_rows(emb, 4), emb[:4])
    finally:
        dist.destroy_process_group()


def _gpu_worker(rank, ws, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=ws)      # two ranks share the box's single GPU: RCCL refuses that
    try:
        import pandas as pd
        from plantcaduceus_amd import embeddings, zero_shot
        from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
        from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
        from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
        cfg = make_config("x", d_model=128, n_layer=2)
        m = CaduceusForMaskedLM(cfg)
        m.load_state_dict(synthetic_state_dict(cfg, seed=5), strict=False)
        m.tie_weights()
        m = m.to("cuda:0")
        df = pd.read_csv(os.path.join(os.path.dirname(__file__), "golden", "example_snp.tsv"), delimiter="\t")
        df = df[df["ref"].isin(list("ACGT")) & df["alt"].isin(list("ACGT"))]
        seqs = df["sequences"].tolist()                                # 185 rows: not divisible by 2
        tok = CaduceusTokenizer()
        p = zero_shot.extract_logits(m, seqs, "cuda:0", 255, tok, batch_size=64)
        e = embeddings.extract_embeddings(m, seqs, "cuda:0", 255, tok, batch_size=64)
        np.savez(os.path.join(outdir, f"g{rank}.npz"), p=p, e=e)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_processes_through_the_hip_engine_equal_single_process(tmp_path):
    """the N > 1 product path end to end on the GPU box — two processes (one process per rank, both on the box's one GPU, gloo
    rendezvous), each running ITS block of the 185 example windows through the HIP engine, one all-gather — equals the
    single-process run bit for bit (windows are independent; rows come back in input order)."""
    mp.spawn(_gpu_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npz"), np.load(tmp_path / "g1.npz")
    assert g0["p"].shape == (185, 4) and g0["e"].shape == (185, 128)
    np.testing.assert_array_equal(g0["p"], g1["p"])
    np.testing.assert_array_equal(g0["e"], g1["e"])
    import pandas as pd
    from plantcaduceus_amd import embeddings, zero_shot
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.modeling_caduceus import CaduceusForMaskedLM
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    cfg = make_config("x", d_model=128, n_layer=2)
    m = CaduceusForMaskedLM(cfg)
    m.load_state_dict(synthetic_state_dict(cfg, seed=5), strict=False)
    m.tie_weights()
    m = m.to("cuda:0")
    df = pd.read_csv(os.path.join(os.path.dirname(__file__), "golden", "example_snp.tsv"), delimiter="\t")
    df = df[df["ref"].isin(list("ACGT")) & df["alt"].isin(list("ACGT"))]
    tok = CaduceusTokenizer()
    np.testing.assert_array_equal(zero_shot.extract_logits(m, df["sequences"].tolist(), "cuda:0", 255, tok, batch_size=64), g0["p"])
    np.testing.assert_array_equal(embeddings.extract_embeddings(m, df["sequences"].tolist(), "cuda:0", 255, tok, batch_size=64), g0["e"])


class _FlagModel:
    """stand-in with the engine's deferred-validation surface: only `bad_rank` saw an invalid token id"""

    def __init__(self, bits):
        self.bits = bits

    def status_bits(self):
        return self.bits

    def check_status(self, bits=None):
        bits = self.bits if bits is None else bits
        if bits:
            raise IndexError(f"status bits {bits}")


def _status_worker(rank, ws, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from plantcaduceus_amd import zero_shot
        zero_shot.check_model_inputs(_FlagModel(0))                 # clean on every rank: nobody raises
        raised = ""
        try:
            zero_shot.check_model_inputs(_FlagModel(1 if rank == 1 else 0))
        except IndexError as ex:
            raised = str(ex)
        t = torch.ones(1)
        dist.all_reduce(t)                                          # the next collective still lines up on every rank
        with open(os.path.join(outdir, f"s{rank}.txt"), "w") as f:
            f.write(raised)
    finally:
        dist.destroy_process_group()


def test_deferred_input_error_raises_on_every_rank(tmp_path):
    """a token-id error seen by ONE rank's shard must surface on all ranks together (the status bits are reduced over
    the group before raising), otherwise the clean ranks walk on into the next collective and hang there."""
    port = _free_port()
    mp.spawn(_status_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "status bits 1" in (tmp_path / f"s{r}.txt").read_text()


def _train_cli_worker(rank, ws, port, outdir, tables):
    """one rank of a `torchrun`-style launch of `python -m plantcaduceus_amd.xgb_train` (env rendezvous, gloo, -device cpu)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    from oracle import caduceus_oracle as O
    from plantcaduceus_amd import xgb_train, zero_shot
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    from plantcaduceus_amd.tokenization_caduceus import CaduceusTokenizer
    torch.set_num_threads(2)
    cfg = make_config("x", d_model=32, n_layer=1)
    model = O.OracleForMaskedLM(O.params_from_state_dict(synthetic_state_dict(cfg, seed=1), cfg))
    model.config = cfg
    zero_shot.load_model_and_tokenizer = lambda d, dev: (model, CaduceusTokenizer())
    xgb_train.main(["-test", tables, "-model", "unused", "-output", outdir, "-device", "cpu", "-tokenIdx", "11", "-seed", "3",
                    "-test_only", "-save_memory", "-chunk_size", "4"])
    assert not dist.is_initialized()                       # every rank left the group (rank 0 before its host-side writing)
    open(os.path.join(outdir, f"done_{rank}"), "w").close()


def test_train_cli_two_ranks_equal_single_process(tmp_path):
    """the train / test command under a 2-rank launch: each chunk's embeddings come from both ranks' blocks through one all-gather,
    only rank 0 writes, both ranks return; files equal the single-process run's."""
    import json
    import pandas as pd
    from plantcaduceus_amd import xgb_predict
    rng = np.random.default_rng(5)
    table = tmp_path / "te.tsv"
    pd.DataFrame({"sequences": ["".join(rng.choice(list("ACGT"), size=24)) for _ in range(10)],
                  "label": [0, 1] * 5}).to_csv(table, sep="\t", index=False)
    tree = dict(left_children=[1, -1, -1], right_children=[2, -1, -1], split_indices=[5, 0, 0], split_conditions=[0.0, -1.0, 1.5],
                default_left=[0, 0, 0], base_weights=[0.0] * 3, parents=[2147483647, 0, 0])
    model_json = {"learner": {"objective": {"name": "binary:logistic"},
                              "learner_model_param": {"base_score": "5E-1", "num_class": "0", "num_feature": "32"},
                              "gradient_booster": {"name": "gbtree", "model": {"trees": [tree], "tree_info": [0]}}}, "version": [2, 0, 3]}
    outs = {}
    for name, ws in (("two", 2), ("one", 1)):
        out = tmp_path / name
        os.makedirs(out)
        json.dump(model_json, open(out / "seed_3_XGBoost.json", "w"))
        mp.spawn(_train_cli_worker, args=(ws, _free_port(), str(out), str(table)), nprocs=ws, join=True)
        assert all((out / f"done_{r}").exists() for r in range(ws))
        outs[name] = out
    for f in ("te_chunk_0_embeddings.npz", "te_chunk_4_embeddings.npz", "te_chunk_8_embeddings.npz"):
        np.testing.assert_array_equal(np.load(outs["two"] / f)["test"], np.load(outs["one"] / f)["test"])
    p2, p1 = (np.load(outs[k] / "seed_3_te_predictions.npz")["predictions"] for k in ("two", "one"))
    np.testing.assert_array_equal(p2, p1)
    assert p2.shape == (10,) and open(outs["two"] / "seed_3_te_metrics.txt").read() == open(outs["one"] / "seed_3_te_metrics.txt").read()
    clf = xgb_predict.XGBJsonClassifier().load_model(str(outs["one"] / "seed_3_XGBoost.json"))
    emb = np.concatenate([np.load(outs["one"] / f"te_chunk_{i}_embeddings.npz")["test"] for i in (0, 4, 8)])
    np.testing.assert_allclose(p1, clf.predict_proba(emb)[:, 1], rtol=1e-6)
