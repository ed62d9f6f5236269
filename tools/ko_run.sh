#!/bin/bash
# same-box A/B of library variants on the attention backward (or forward with AB_FWD=1):  tools/ko_run.sh "N L" lib1 lib2 ...
SHAPE=${1:-"16 4096"}; shift
for r in 1 2; do
for lib in visionselector_amd/libvsel.so "$@"; do
  echo "== $lib"
  if [ "${AB_FWD:-0}" = 1 ]; then python tools/bench_attn.py --shape $SHAPE --lib $lib 2>&1 | grep n_seq
  else python tools/bench_attn_bwd.py --shape $SHAPE --lib $lib 2>&1 | grep n_seq; fi
done; done
