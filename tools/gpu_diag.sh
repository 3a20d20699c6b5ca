#!/bin/bash
# The diagnostic builds behind DESIGN section 4's stamp / ablation figures, in one GPU session.  Usage: tools/gpu_diag.sh TAG
# -> gpurun_out/TAG/{stamps_nt256_pair.log, stamps_nt160lw.log, stamps_nt160p.log, tn3lw_ablation.log, nt_variants_*.log}
set -o pipefail
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
HC="hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I clip_event_amd/csrc -I include"
$HC tools/diag/nt256_stamps.hip -o /tmp/nt256_stamps || exit 1
for k in 768 3072; do /tmp/nt256_stamps $k 1 0; /tmp/nt256_stamps $k 0 0; done > $OUT/stamps_nt256_pair.log 2>&1
python tools/diag/make_nt160lw_stamps.py /tmp/lw.hip && $HC /tmp/lw.hip -o /tmp/lw || exit 1
for k in 768 3072; do /tmp/lw $k 1 0; /tmp/lw $k 0 0; /tmp/lw $k 0 2; done > $OUT/stamps_nt160lw.log 2>&1
for e in CE_EPI_BF16 CE_EPI_BIAS_GELU CE_EPI_GELUGRAD_BF16; do
  python tools/diag/make_nt160p_stamps.py /tmp/p.hip $e && $HC /tmp/p.hip -o /tmp/p || exit 1
  echo "== $e"; /tmp/p 768 0; /tmp/p 768 1
done > $OUT/stamps_nt160p.log 2>&1
bash tools/diag/tn3_ablate.sh > $OUT/tn3lw_ablation.log 2>&1 || exit 1
for e in 0 5 6; do
  echo "== EPI=$e"; EPI=$e VARIANTS=0,104,5,160,161,32,162 python tools/diag/nt_variants.py 2>/dev/null
done > $OUT/nt_variants.log
tail -3 $OUT/stamps_nt160lw.log | cut -c1-200
