#!/bin/bash
# round 5, GPU call 1 (developer script, run through gpurun): census oracle runs on the box's host cores in the background,
# new GPU tests + bench lines of the three operation orders meanwhile, then the census test / report from the fresh fixtures.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r05a; mkdir -p $O
export OMP_WAIT_POLICY=passive HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "from oracle import c_oracle; c_oracle.build(); print('oracle built, threads', c_oracle.COracle.__name__)" > $O/build.log 2>&1
nproc >> $O/build.log
( OMP_NUM_THREADS=${CENSUS_THREADS:-96} python oracle/gen_census_golden.py --model l32 --n 512 --n-eng 256 --n-plainc 32 --out $O/census_l32.npz > $O/census_l32.log 2>&1
  OMP_NUM_THREADS=${CENSUS_THREADS:-96} python oracle/gen_census_golden.py --model l20 --n 256 --n-plainc 32 --out $O/census_l20.npz > $O/census_l20.log 2>&1 ) &
CPID=$!
# new / touched tests first (few host threads: the census owns the cores)
OMP_NUM_THREADS=24 timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_dist.py tests/test_gpu_normfold.py -m gpu -q -x -s > $O/tests_new.log 2>&1
echo "tests_new rc=$?" >> $O/status.txt
# the three operation orders, same box, interleaved twice (GPU-only: no CPU baseline / host timings)
for r in 1 2; do
  for o in "" "--opt reference_order=1" "--opt reference_order=2"; do
    timeout 300 python bench.py --steps 8 --warmup 3 --cpu-seqs 0 --host-seqs 0 $o > $O/bench_l32_r${r}_$(echo "$o" | tr -c 'a-z0-9=_' '_').json 2>> $O/bench.err
  done
done
for o in "" "--opt reference_order=1" "--opt reference_order=2"; do
  timeout 300 python bench.py --model l20 --steps 8 --warmup 3 --cpu-seqs 0 --host-seqs 0 $o > $O/bench_l20_$(echo "$o" | tr -c 'a-z0-9=_' '_').json 2>> $O/bench.err
done
echo "bench done" >> $O/status.txt
# the rest of the GPU suite while the census still runs (its oracle-heavy tests share the cores)
OMP_NUM_THREADS=48 timeout 3000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_census.py > $O/tests_all.log 2>&1
echo "tests_all rc=$?" >> $O/status.txt
wait $CPID
echo "census done" >> $O/status.txt
cp $O/census_l32.npz $O/census_l20.npz tests/golden/ 2>> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_census.py -m gpu -q -s > $O/tests_census.log 2>&1
echo "tests_census rc=$?" >> $O/status.txt
timeout 600 python tools/argmax_census.py --model l32 --fixture tests/golden/census_l32.npz > $O/argmax_census_l32.txt 2>&1
timeout 600 python tools/argmax_census.py --model l20 --fixture tests/golden/census_l20.npz --batch 1024 > $O/argmax_census_l20.txt 2>&1
# default bench line on a quiet host (cpu_baseline, host timings, energy)
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "all done" >> $O/status.txt
