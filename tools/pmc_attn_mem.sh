#!/bin/bash
# Memory-side counters of the attention kernels (forward inside run_attn_bwd.py + both backward kernels) at N_SEQ x L:
# L2 hits / misses and memory-side read requests, one rocprofv3 --pmc pass each.   tools/pmc_attn_mem.sh 16 4096 out_dir
set -u
ROOT="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
NS=${1:-16}; L=${2:-4096}; OUT=$ROOT/gpurun_out/${3:-pmc_attn_mem}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for P in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $ROOT/tools/run_attn_bwd.py $NS $L 1 > /dev/null 2> $OUT/p$i.err
done
python $ROOT/tools/pmc_summary.py $OUT/p1 $OUT/p2 > $OUT/summary.txt
find $OUT -name '*.csv' -size +4M -delete
cat $OUT/summary.txt
