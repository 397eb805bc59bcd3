cd $GRAFT_REPO_ROOT; O=gpurun_out
for i in 1 2; do timeout 200 python bench.py --prompt-tokens 32 --no-sft --no-sustain --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('S=289 after: tok/s',d['value'],'ttft',d['ttft_ms'],'tower',d['prefill']['encode_images_ms'])"; done
timeout 200 python bench.py --no-sft --no-sustain --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('S=769: tok/s',d['value'],'ttft',d['ttft_ms'])"
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py tests/test_gpu_full_size.py tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -4
