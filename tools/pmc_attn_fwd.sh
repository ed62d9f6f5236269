#!/bin/bash
# PMC counters of the attention forward kernel at N_SEQ x L (separate passes of <= 8 SQ counters): tools/pmc_attn_fwd.sh 16 4096 out_dir
set -u
ROOT="${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
NS=${1:-16}; L=${2:-4096}; OUT=$ROOT/gpurun_out/${3:-pmc_fwd}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $ROOT/tools/run_attn_batch.py $NS $L 1 5 > /dev/null 2> $OUT/p$i.err
done
python $ROOT/tools/pmc_summary.py $OUT/p1 $OUT/p2 > $OUT/summary.txt
find $OUT -name '*.csv' -size +4M -delete
cat $OUT/summary.txt
