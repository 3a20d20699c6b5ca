#!/bin/bash
# Ablation builds of gemm_tn3lw_kernel (separate .so under /tmp, never the library): 0 = as shipped, 1 = no epilogue
# atomics, 2 = no LDS-DMA (barriers and MFMAs only).  Prints tools/bench_tn_group.py per build.  Needs the library built.
set -e
cd "$(dirname "$0")/../.."
for mode in 0 1 2; do
  hipcc -O3 -std=c++17 -fPIC -munsafe-fp-atomics -w --offload-arch=gfx950 -DCE_DIAG_TN3=$mode -x hip -c clip_event_amd/csrc/gemm.hip -o /tmp/gemm_diag_$mode.o
  objs=$(ls clip_event_amd/build/*.o | grep -v "/gemm.hip.o")
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libce_diag_$mode.so /tmp/gemm_diag_$mode.o $objs
  echo "== CE_DIAG_TN3=$mode"
  CE_DIAG_LIB=/tmp/libce_diag_$mode.so python tools/bench_tn_group.py 2>/dev/null
done
