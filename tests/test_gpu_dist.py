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


def _torchrun(args, port, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
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
