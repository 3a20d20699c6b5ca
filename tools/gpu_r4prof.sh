#!/bin/bash
# round 4: the judged artefacts -- bench line (+ roofline, cpu baseline, power sample), rocprofv3 kernel stats of the same
# command on one stream, PMC passes (HBM traffic, MFMA utilisation, SQ wait states), idle accounting, config 3 / 4 / 5 lines
set -o pipefail
TAG=${1:-r4prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
cut -c1-200 $OUT/bench.json
tools/gpu_prof.sh $TAG > $OUT/stats_summary.txt || exit 1
head -3 $OUT/stats_summary.txt
BARGS="--single-stream --no-cpu-baseline --no-roofline --no-dense-compare --steps 3 --warmup 1"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq --output-format csv -- python3 $ROOT/bench.py $BARGS > $OUT/pmc_sq.log 2>&1 || exit 1
# config 5 geometry: MFMA / LDS counters of the long-sequence attention kernels
C5ARGS="--arch vit_l14_336 --single-stream --no-cpu-baseline --no-roofline --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma_c5 --output-format csv -- python3 $ROOT/bench.py $C5ARGS > $OUT/pmc_mfma_c5.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq_c5 --output-format csv -- python3 $ROOT/bench.py $C5ARGS > $OUT/pmc_sq_c5.log 2>&1 || exit 1
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_hbm_traffic.json
python3 tools/pmc_mfma.py $OUT/pmc_mfma > $OUT/pmc_mfma_util.json
python3 tools/pmc_generic.py $OUT/pmc_sq 16 > $OUT/pmc_sq.txt
python3 tools/pmc_mfma.py $OUT/pmc_mfma_c5 > $OUT/pmc_mfma_util_config5.json
python3 tools/pmc_generic.py $OUT/pmc_sq_c5 12 > $OUT/pmc_sq_config5.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_sq $OUT/pmc_mfma_c5 $OUT/pmc_sq_c5
python bench.py --batch 64 --descriptions 5 --alignment --train-arg desc --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_config4.json 2> $OUT/bench_c4.err || exit 1
python bench.py --arch vit_l14_336 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_config5_bf16.json 2> $OUT/bench_c5.err || exit 1
python bench.py --arch vit_l14_336 --fp8 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_config5_fp8.json 2> $OUT/bench_c5f.err || exit 1
python bench.py --descriptions 5 --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $OUT/bench_config3_shape.json 2> $OUT/bench_c3.err || exit 1
python bench.py --batch 512 --descriptions 5 --no-cpu-baseline --no-roofline --steps 5 --warmup 2 > $OUT/bench_config3_per_rank.json 2> $OUT/bench_c3r.err || exit 1
for f in bench_config4 bench_config5_bf16 bench_config5_fp8 bench_config3_shape bench_config3_per_rank; do python -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', d['ms_per_step'], 'ms', d['value'], 'pairs/s', (d['roofline'] or {}).get('step_frac'))"; done
# idle accounting
cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-roofline --no-dense-compare --steps 10 --warmup 3 > $OUT/trace_bench.log 2>&1 || exit 1
cd $ROOT
python3 tools/trace_idle.py $OUT/trace $OUT/timeline_two_stream.txt > $OUT/trace_idle_two_stream.txt
head -3 $OUT/trace_idle_two_stream.txt
rm -rf $OUT/trace
