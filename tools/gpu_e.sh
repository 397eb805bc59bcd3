#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_w8a8.py tests/test_gpu_serving.py tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_video_encoders.py tests/test_gpu_edge.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r2e_pytest.log
tail -15 gpurun_out/r2e_pytest.log
timeout 300 python bench.py --w8-vit --no-cpu-baseline --no-sft 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w8 vit: ttft', d['ttft_ms'], 'encode', d['prefill']['encode_images_ms'], 'tok/s', d['value'])"
timeout 300 python bench.py --no-cpu-baseline --no-sft 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 vit: ttft', d['ttft_ms'], 'encode', d['prefill']['encode_images_ms'], 'tok/s', d['value'])"
