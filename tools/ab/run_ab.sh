#!/bin/bash
# same-box A/B of two builds of libvila_hip.so (tools/ab/libvila_hip_{old,new}.so, git-ignored): alternating runs
#   gpurun --timeout 1200 -- bash tools/ab/run_ab.sh          results under gpurun_out/ab/
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab; mkdir -p $O
use() { cp tools/ab/libvila_hip_$1.so vila_amd/lib/libvila_hip.so; }
for rep in 1 2; do for v in old new; do
  use $v
  timeout 200 tools/gemm_bench one 7 2>&1 | grep "sched=0" | sed "s/^/$v /" >> $O/gemm_one.log
  timeout 300 python bench.py --mode sft --steps 4 --warmup 2 2>$O/sft_$v$rep.err | tail -1 > $O/sft_$v$rep.json
  python -c "import json; d=json.loads(open('$O/sft_$v$rep.json').read()); print('$v run $rep: sft', d['ms_per_step'], 'ms')"
done; done
for v in old new old new; do
  use $v
  timeout 300 python bench.py --no-sft --no-sustain --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | tail -1 > $O/bench_$v.json
  python -c "import json; d=json.loads(open('$O/bench_$v.json').read()); print('$v: decode', d['value'], 'ttft', d['ttft_ms'])"
done
use new
cat $O/gemm_one.log
