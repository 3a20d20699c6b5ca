#!/bin/bash
# A/B of a compile-time switch of one csrc file in the step: tools/diag/ab_macro.sh FILE.hip "-DMACRO=0" "-DMACRO=1" ...
# builds /tmp/libce_ab_N.so per flag set (other objects from clip_event_amd/build) and runs the bench against each, twice.
f=$1; shift
base=$(basename $f)
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc -O3 -std=c++17 -fPIC -munsafe-fp-atomics -w --offload-arch=gfx950 $flags -x hip -c clip_event_amd/csrc/$f -o /tmp/ab_$i.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libce_ab_$i.so /tmp/ab_$i.o $(ls clip_event_amd/build/*.o | grep -v "/$base.o") || exit 1
done
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    CE_DIAG_LIB=/tmp/libce_ab_$i.so python tools/diag/bench_with_lib.py --steps 15 --warmup 4 --no-cpu-baseline --no-dense-compare $BENCH_ARGS > /tmp/ab_$i.json 2>/tmp/ab_$i.err || { tail -3 /tmp/ab_$i.err; exit 1; }
    python - "$flags" /tmp/ab_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
cl = d.get("roofline", {}).get("classes", [])
nt = sum(c["ms_per_step"] for c in cl if "gemm_nt" in c["kernel"])
print(f"{sys.argv[1]:28s} {d['ms_per_step']:7.3f} ms/step   NT classes {nt:.3f} ms   all classes {sum(c['ms_per_step'] for c in cl):.2f} ms")
PY
  done
done
