"""developer tool: where the CPU baseline (oracle/c, the port bench.py times) spends its time on this host — per-operator wall
clock of one l32 forward (ORACLE_TIMING=1 makes oracle_forward print it), fp32 and bf16-emulating.   python tools/oracle_timing.py [windows]"""
import os, sys, time
os.environ["ORACLE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.c_oracle import COracle
from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = make_config("l32")
sd = synthetic_state_dict(cfg, seed=1234, stress=False)
ids = np.random.default_rng(0).integers(3, 7, size=(n, 512)).astype(np.int32)
for name, kw in (("fp32, host BLAS", dict(blas=True)), ("fp32, plain-C GEMM", dict()),
                 ("bf16-emulating (reference order), host BLAS", dict(blas=True, dtype=torch.bfloat16, emulate_bf16=True, ref_order=True))):
    o = COracle(sd, cfg, **kw)
    t = time.time()
    o.forward(ids)
    dt = time.time() - t
    print(f"{name}: {n} windows in {dt:.1f} s = {n / dt:.2f} seq/s on {o.threads} threads", flush=True)
