#!/usr/bin/env python3
"""developer tool: what the GPU box's HOST offers the CPU baseline (bench.py `cpu_baseline`, oracle/c) - CPU quota / affinity, and the
per-operator time of the port with its projections through numpy's sgemm, torch.mm or the plain-C loops, at several team sizes and
with / without passive waiting (OpenBLAS workers and OpenMP teams that spin after their region burn a CFS quota).
    python tools/host_probe.py [windows]           (CPU only; writes nothing, prints)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "16"

if len(sys.argv) > 2:            # child: one variant
    os.environ["ORACLE_TIMING"] = "1"
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    from oracle.c_oracle import COracle
    from plantcaduceus_amd.checkpoint import make_config, synthetic_state_dict
    cfg = make_config("l32")
    sd = synthetic_state_dict(cfg, seed=1234, stress=False)
    blas = {"plain": False, "numpy": "numpy", "torch": "torch"}[sys.argv[2]]
    ids = np.random.default_rng(0).integers(3, 7, size=(int(n), 512)).astype(np.int32)
    o = COracle(sd, cfg, blas=blas, threads=int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) else None)
    o.forward(ids[:1])
    t = time.time()
    o.forward(ids)
    dt = time.time() - t
    print("   -> %s: %s windows in %.2f s = %.2f seq/s, OpenMP team %d" % (sys.argv[2], n, dt, int(n) / dt, o.threads), flush=True)
    sys.exit(0)


def sh(cmd):
    r = subprocess.run(cmd, shell=True, capture_output=True, text=True)
    print("$ %s\n%s" % (cmd, (r.stdout + r.stderr).strip()), flush=True)


sh("cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; nproc; cat /proc/loadavg")
sh("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node|MHz|L3' | head -14")
sh("python3 -c \"import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())\"")
sh("python3 -c \"import numpy, threadpoolctl, torch, json; print(json.dumps(threadpoolctl.threadpool_info())[:1500]); print('torch threads', torch.get_num_threads(), torch.__config__.parallel_info()[:400])\"")
PASSIVE = {"OPENBLAS_THREAD_TIMEOUT": "4", "OMP_WAIT_POLICY": "passive"}
for env, variant, threads in ((PASSIVE, "numpy", 16), ({}, "numpy", 16), (PASSIVE, "torch", 16), (PASSIVE, "numpy", 32), (PASSIVE, "plain", 16),
                              (PASSIVE, "numpy", 0), ({}, "numpy", 0))[:int(os.environ.get("PROBE_VARIANTS", "7"))]:
    print("== %s threads=%s %s" % (variant, threads or "default", env), flush=True)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), n, variant, str(threads)], env=dict(os.environ, **env), capture_output=True, text=True)
    print("\n".join(ln for ln in (r.stdout + r.stderr).splitlines() if "oracle_forward" in ln or "->" in ln or "Error" in ln), flush=True)
