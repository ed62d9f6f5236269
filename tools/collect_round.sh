#!/bin/bash
# Round evidence on one MI355X box: bench JSON, rocprofv3 kernel stats of the same command, PMC traffic (separate FETCH_SIZE /
# WRITE_SIZE passes), MFMA counters of the LIS projections, the batch table.  Everything lands in gpurun_out/$1/ ;
# copy what is to be judged into profiles/.
R=${1:-r04}
set -u
ROOT="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is set there)}"
cd "$ROOT"
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-train --no-attn --no-llm --no-single-sweep --no-batch1 --no-configs"
# 1. the default bench line (all legs)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# 2. rocprofv3 --kernel-trace --stats of the headline launches
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b128 -o lis -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/prof_b128.err)
python tools/summarize_rocprof.py $(find $OUT/prof_b128 -name '*kernel_stats.csv' | head -1) $OUT/bench_b128_kernel_stats.csv > /dev/null
python tools/trace_gaps.py $(find $OUT/prof_b128 -name '*kernel_trace.csv' | head -1) 8 > $OUT/bench_b128_gaps.txt
find $OUT/prof_b128 -name '*kernel_trace.csv' -delete
# 2b. the same with the single-sweep leg: gelu_colsum / colsum_linear / presummed select in the kernel stats
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ss -o lis -- python $ROOT/bench.py --no-cpu-baseline --no-train --no-attn --no-llm --no-batch1 --no-configs --steps 10 --warmup 3 > $OUT/bench_single_sweep_under_rocprof.json 2> $OUT/prof_ss.err)
python tools/summarize_rocprof.py $(find $OUT/prof_ss -name '*kernel_stats.csv' | head -1) $OUT/bench_b128_single_sweep_kernel_stats.csv > /dev/null
find $OUT/prof_ss -name '*kernel_trace.csv' -delete
# 3. HBM traffic: separate PMC passes
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- $BENCH --steps 5 --warmup 2 > /dev/null 2> $OUT/pmc_$C.err)
done
python tools/pmc_to_json.py $(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) 128 $OUT/pmc_traffic.json > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc1_$C -o p -- $BENCH --images 1 --steps 20 --warmup 2 > /dev/null 2> $OUT/pmc1_$C.err)
done
python tools/pmc_to_json.py $(find $OUT/pmc1_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $OUT/pmc1_WRITE_SIZE -name '*counter_collection.csv' | head -1) 1 $OUT/pmc_traffic.json > /dev/null
# 4. MFMA utilisation of the scorer's dense projections (north_star: "MFMA used only for the scorer's dense projection")
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_mfma -o p -- python $ROOT/tools/run_lis.py 128 10 > /dev/null 2> $OUT/pmc_mfma.err)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_mfma1 -o p -- python $ROOT/tools/run_lis.py 1 50 > /dev/null 2> $OUT/pmc_mfma1.err)
python tools/mfma_summary.py $OUT/pmc_mfma $OUT/pmc_mfma1 > $OUT/lis_mfma.json
# 5. batch table (driver-style bench at B = 1, 8, 32, 128)
for B in 1 8 32 128; do $BENCH --images $B --steps 50 --warmup 5; done > $OUT/batch_table.jsonl 2> $OUT/batch_table.err
find $OUT -name '*counter_collection.csv' -size +8M -delete
ls -la $OUT
