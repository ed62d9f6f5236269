#!/bin/bash
# Round 5: rocprofv3 kernel stats + PMC passes (SQ utilisation, LDS, L2) of the attention kernels that carry the driver numbers:
# attn_fwd64 / attn_bwd_dq64 / attn_bwd_dkdv64 at 16 x 4096 and attn_fwd_gqa at 32 x 524 (7B heads).  gpurun_out/r05_attn/ -> profiles/r05_attn_*.
set -u
ROOT="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$ROOT"
O=$ROOT/gpurun_out/r05_attn
mkdir -p $O; export TMPDIR=/tmp
# kernel stats (durations) of the same commands
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_bwd -o s -- python $ROOT/tools/run_attn_bwd.py 16 4096 5 > /dev/null 2> $O/st_bwd.err)
python tools/summarize_rocprof.py $(find $O/st_bwd -name '*kernel_stats.csv' | head -1) $O/kernel_stats_16x4096_fwd_bwd.csv > /dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_gqa -o s -- python $ROOT/tools/run_attn_batch.py 32 524 1 20 > /dev/null 2> $O/st_gqa.err)
python tools/summarize_rocprof.py $(find $O/st_gqa -name '*kernel_stats.csv' | head -1) $O/kernel_stats_32x524_fwd.csv > /dev/null
find $O -name '*kernel_trace.csv' -delete
# SQ counters (two passes of 8) + memory side (two passes)
timeout 900 bash tools/pmc_attn_fwd.sh 16 4096 r05_attn/pmc_fwd64 > /dev/null 2>&1
timeout 900 bash tools/pmc_attn_bwd.sh 16 4096 r05_attn/pmc_bwd64 > /dev/null 2>&1
timeout 900 bash tools/pmc_attn_fwd.sh 32 524 r05_attn/pmc_gqa > /dev/null 2>&1
timeout 900 bash tools/pmc_attn_mem.sh 16 4096 r05_attn/pmc_mem64 > /dev/null 2>&1
i=0
for P in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_mem_gqa/p$i -o p -- python $ROOT/tools/run_attn_batch.py 32 524 1 5 > /dev/null 2> $O/pmc_mem_gqa_p$i.err)
done
python tools/pmc_summary.py $O/pmc_mem_gqa/p1 $O/pmc_mem_gqa/p2 > $O/pmc_mem_gqa/summary.txt
for d in pmc_fwd64 pmc_bwd64 pmc_gqa pmc_mem64 pmc_mem_gqa; do echo "==== $d"; cat $O/$d/summary.txt; done > $O/pmc_all.txt
find $O -name '*.csv' -size +2M -delete
cat $O/kernel_stats_16x4096_fwd_bwd.csv $O/kernel_stats_32x524_fwd.csv; cat $O/pmc_all.txt
