#!/bin/bash
set -o pipefail
TAG=${1:-r4f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -30
bash tools/gpu_ab_env.sh $TAG 3 "CE_GEMM_CUS=256 --" "CE_GEMM_CUS=248 --" "CE_GEMM_CUS=240 --" "CE_GEMM_CUS=232 --"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-compare 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['power_sample'])"
