#!/bin/bash
# Same-box A/B of two builds of libvsel.so on the large attention-forward shapes (boxes differ by +-3 %, so two builds are only
# comparable inside one gpurun call):  [AB_BWD=1] tools/ab_attn.sh A.so B.so [rounds]   -> alternating runs, TFLOP/s per shape and run
# (AB_BWD=1: the backward, tools/bench_attn_bwd.py)
A=$1; B=$2; R=${3:-3}; AB_ARGS=${AB_ARGS---big}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    echo "== $lib (round $r)"
    if [ "${AB_BWD:-0}" = 1 ]; then python tools/bench_attn_bwd.py $AB_ARGS --lib $lib 2>&1 | grep n_seq
    else python tools/bench_attn.py $AB_ARGS --lib $lib 2>&1 | grep n_seq | sed "s/'us': [0-9.]*, //"; fi
  done
done
