#!/bin/bash
# Collect the judged artefacts of a round on the GPU box: bench line, rocprofv3 kernel stats, PMC passes.
# Usage: tools/gpu_profiles.sh TAG   -> gpurun_out/TAG/{bench.json, stats.csv, bench_single_under_rocprof.json,
#                                        pmc_hbm_traffic.json, pmc_mfma_util.json, bench_c4.json}
set -o pipefail
TAG=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
cut -c1-300 $OUT/bench.json
tools/gpu_prof.sh $TAG > $OUT/stats_summary.txt || exit 1
head -3 $OUT/stats_summary.txt
BARGS="--single-stream --no-cpu-baseline --no-roofline --no-dense-compare --steps 3 --warmup 1"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_sq.log 2>&1 || exit 1
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_hbm_traffic.json
python3 tools/pmc_mfma.py $OUT/pmc_mfma > $OUT/pmc_mfma_util.json
python3 tools/pmc_generic.py $OUT/pmc_sq 16 > $OUT/pmc_sq.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_sq
python bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err || exit 1
cut -c1-260 $OUT/bench_c4.json
head -c 700 $OUT/pmc_hbm_traffic.json
