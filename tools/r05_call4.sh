#!/bin/bash
# round 5, GPU call 4 (developer script): f32_gemm_split with the scan-written out_proj operand and the split dt_proj - tests, then the fp32 model's bench lines
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "f32_gemm_split or forward_fp32 or segmented or plantcad2" > $O/tests_model.log 2>&1; echo "model rc=$?" >> $O/status.txt
timeout 2400 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q -x -s -k "fp32" > $O/tests_fulldepth.log 2>&1; echo "fulldepth rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "scan or conv_xproj" > $O/tests_ops.log 2>&1; echo "ops rc=$?" >> $O/status.txt
B="--dtype f32 --steps 4 --warmup 2 --cpu-seqs 0 --host-seqs 0"
for r in 1 2; do
  timeout 600 python bench.py $B > $O/bench_f32_plain_r$r.json 2>> $O/bench.err
  timeout 600 python bench.py $B --opt f32_gemm_split=1 > $O/bench_f32_split_r$r.json 2>> $O/bench.err
done
timeout 600 python bench.py --model l20 $B --opt f32_gemm_split=1 > $O/bench_l20_f32_split.json 2>> $O/bench.err
timeout 600 python bench.py --model pc2-large --batch 512 $B --opt f32_gemm_split=1 > $O/bench_pc2large_f32_split.json 2>> $O/bench.err
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r05d/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"], 1), "seq/s", round(d["ms_per_step"], 1), "ms", {k: v["avg_ms"] for k, v in d.get("kernels", {}).items()})
    except Exception as ex:
        print(f, "FAILED", ex)
PY
echo "all done" >> $O/status.txt
