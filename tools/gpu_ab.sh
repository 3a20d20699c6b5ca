#!/bin/bash
# A/B of env-selected variants in one GPU session (tn column: the gemm_tn3_kernel class): tools/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV..." ...
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-dense-compare > gpurun_out/$TAG/ab_$i.json 2> gpurun_out/$TAG/ab_$i.err || { echo "variant $i failed"; tail -5 gpurun_out/$TAG/ab_$i.err; exit 1; }
  python - "$envs" gpurun_out/$TAG/ab_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
r = d["roofline"]
cls = {c["kernel"]: c for c in r["classes"]}
tn = cls.get("gemm_tn3lw_kernel", cls.get("gemm_tn3_kernel", cls.get("gemm_tn", {})))
print(f"{sys.argv[1]:40s} {d['ms_per_step']:7.3f} ms/step  tn {tn.get('ms_per_step', 0):.3f} ms ({tn.get('tflops', 0):.0f} TF, {tn.get('launches_per_step', 0):.0f} launches)  "
      f"sum_classes {sum(c['ms_per_step'] for c in r['classes']):.2f} ms")
PY
done
