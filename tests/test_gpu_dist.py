"""-m gpu: the distributed branch (RCCL process group, all_gather_into_tensor, barrier, all_reduce) executed on the one GPU of
the test box — what `--gpus 8` / an 8-rank torchrun run, at world size 1, each as a child process started with torchrun."""
import json
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port, timeout=600, nproc=1):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_distributed_branch_world1():
    """bench.py exactly as the driver launches it for N > 1 (torchrun, RANK / WORLD_SIZE / MASTER_* from the environment), with
    --force-dist so the nccl group, the in-loop all-gather, the barriers and the MAX all-reduce run at world size 1."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--batch", "64",
                   "--model", "l20", "--cpu-seqs", "0", "--host-seqs", "0"], _free_port())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["distributed_branch_executed"] is True and d["value"] > 0
    assert d["unit"] == "sequences/s" and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["rccl_ranks_seen"] == [0] and len(d["per_rank_seq_per_s"]) == 1


def _visible_gpus():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU on this image


# N > 1 over RCCL: self-verifying the moment a box shows more than one device (the builder's boxes have one GPU: these are
# collected and skipped there; SURVEY.md §8(e), BASELINE config 4).  Children are torchrun ranks, one per device, exactly as the
# driver launches bench.py --gpus N.
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_multi_rank_equals_single_rank(world, tmp_path):
    """`world` ranks on `world` devices: each runs its block of the 185 example windows through the HIP engine, ONE
    all_gather_into_tensor over RCCL/xGMI per result; every rank must hold the full [185, 4] / [185, D] result, bit-identical to
    the one-rank run, and the ranks must report `world` distinct devices on the nccl backend.  Also the sharded PlantCAD2 loops
    (masked_probs at three positions of 37 windows, unmasked_probs of 21 windows; one gather per 3-window chunk).  The worker's
    engine runs with "scan_segments" 0 so that per-call batch sizes (which differ between world 1 and world N) do not select
    different small-launch forms (fp32 summation order)."""
    if _visible_gpus() < world:
        pytest.skip(f"needs {world} GPUs, {_visible_gpus()} visible")
    worker = os.path.join(ROOT, "tests", "_rccl_worker.py")
    d1, dn = tmp_path / "w1", tmp_path / f"w{world}"
    d1.mkdir()
    dn.mkdir()
    r = _torchrun([worker, str(dn)], _free_port(), nproc=world)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r1 = _torchrun([worker, str(d1)], _free_port(), nproc=1)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    ref = np.load(d1 / "r0.npz")
    assert ref["p"].shape == (185, 4) and ref["e"].shape == (185, 128)
    reports = [json.load(open(dn / f"r{k}.json")) for k in range(world)]
    assert sorted(x["rank"] for x in reports) == list(range(world))
    assert all(x["world"] == world and x["backend"] == "nccl" for x in reports)
    assert len({x["device"] for x in reports}) == world, reports            # one device per rank
    assert ref["mp3"].shape == (37 * 3, 4) and ref["un"].shape == (21, 512, 4)
    for k in range(world):
        got = np.load(dn / f"r{k}.npz")
        for key in ("p", "e", "mp3", "un"):
            np.testing.assert_array_equal(got[key], ref[key])


def test_rccl_worker_world1(tmp_path):
    """the same child at one rank on the one GPU at hand, so that the worker the N > 1 tests rely on is itself executed on
    every box (an N > 1 failure is then about N, not about the worker)."""
    r = _torchrun([os.path.join(ROOT, "tests", "_rccl_worker.py"), str(tmp_path)], _free_port(), nproc=1)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rep = json.load(open(tmp_path / "r0.json"))
    assert rep["rank"] == 0 and rep["world"] == 1 and rep["backend"] == "nccl"
    got = np.load(tmp_path / "r0.npz")
    assert got["p"].shape == (185, 4) and got["e"].shape == (185, 128) and np.isfinite(got["p"]).all()
    np.testing.assert_allclose(got["p"].sum(1), 1.0, rtol=1e-5)
    assert got["mp3"].shape == (111, 4) and got["un"].shape == (21, 512, 4)
    np.testing.assert_allclose(got["un"].sum(-1), 1.0, rtol=1e-5)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_rccl_multi_rank(world):
    """bench.py as the driver launches it for N > 1; the JSON line must name N ranks seen through the RCCL group and carry one
    per-rank rate each, and the aggregate must be the sum of the work over the slowest rank's time."""
    if _visible_gpus() < world:
        pytest.skip(f"needs {world} GPUs, {_visible_gpus()} visible")
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "64",
                   "--model", "l20", "--cpu-seqs", "0", "--host-seqs", "0"], _free_port(), nproc=world)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == world and d["rccl_ranks_seen"] == list(range(world)) and len(d["per_rank_seq_per_s"]) == world
    assert d["distributed_branch_executed"] is True and d["scaling"] == "weak"
    assert d["value"] <= sum(d["per_rank_seq_per_s"]) * 1.001 and d["value"] > 0


def test_zero_shot_cli_under_torchrun_world1(golden_dir, tmp_path):
    """the zero-shot command line under a one-rank torchrun: process group over RCCL, this rank's block through the HIP engine,
    all_gather_into_tensor of the [N, 4] probabilities, rank 0 writes — equal to the plain single-process run."""
    from plantcaduceus_amd import zero_shot
    from plantcaduceus_amd.checkpoint import make_synthetic_checkpoint
    d = str(tmp_path / "snap")
    make_synthetic_checkpoint(d, "x", seed=13, stress=False, d_model=128, n_layer=2)
    src = pd.read_csv(os.path.join(golden_dir, "example_snp.tsv"), delimiter="\t").iloc[:40]
    inp, out1, out2 = tmp_path / "in.tsv", tmp_path / "o1.tsv", tmp_path / "o2.tsv"
    src.to_csv(inp, sep="\t", index=False)
    r = _torchrun(["-m", "plantcaduceus_amd.zero_shot", "-input-table", str(inp), "-output", str(out1), "-model", d,
                   "-device", "cuda:0", "-batchSize", "16"], _free_port())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    zero_shot.main(["-input-table", str(inp), "-output", str(out2), "-model", d, "-device", "cuda:0", "-batchSize", "16"])
    a, b = pd.read_csv(out1, delimiter="\t"), pd.read_csv(out2, delimiter="\t")
    assert len(a) == len(b) > 0
    np.testing.assert_array_equal(a["zeroShotScore"].to_numpy(), b["zeroShotScore"].to_numpy())
