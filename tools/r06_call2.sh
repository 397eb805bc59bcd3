#!/bin/bash
# round 6, second GPU call: the persistent decode token kernel — parity with the per-kernel step, then the A/B on the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "persistent or chained" 2>&1 | tail -30 > $O/r06_call2_tests.log
tail -12 $O/r06_call2_tests.log
for i in 1 2; do
  for m in 0 1; do
    VILA_DECODE_PERSIST=$m timeout 300 python bench.py --no-sft --no-cpu-baseline --no-sustain 2>>$O/r06_call2.err | tail -1 > $O/r06_decode_persist${m}_$i.json
    python -c "
import json
d=json.loads(open('$O/r06_decode_persist${m}_$i.json').read()); print('persist=$m run $i: value', d['value'], 'ms/step', d['ms_per_step'], 'ttft', d.get('ttft_ms'), 'launches', d.get('config',{}).get('launches_per_token'))"
  done
done
cd /tmp && export TMPDIR=/tmp
VILA_DECODE_PERSIST=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_persist -- python $GRAFT_REPO_ROOT/bench.py --no-sft --no-cpu-baseline --no-sustain --steps 64 > /dev/null 2>>$GRAFT_REPO_ROOT/$O/r06_call2.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_persist -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
