#!/bin/bash
# round-2 first GPU pass: full GPU suite, default bench line, RCCL path at world 1, TSP video prefill
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -80 > gpurun_out/r2a_pytest.log
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
timeout 600 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?"
VILA_BENCH_FORCE_DIST=1 timeout 600 python bench.py --mode sft > gpurun_out/r2a_sft_forcedist.json 2> gpurun_out/r2a_sft_forcedist.err
echo "sft rc=$?"
timeout 600 python bench.py --mode video --tsp > gpurun_out/r2a_video_tsp.json 2> gpurun_out/r2a_video_tsp.err
echo "video rc=$?"
tail -5 gpurun_out/r2a_pytest.log
cat gpurun_out/r2a_bench.json | cut -c1-1500
