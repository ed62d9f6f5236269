#!/bin/bash
# every trace build under visionselector_amd/build/variants (tools/ab_fwd64.py build name:trace=1,...): one line of cycle accounting each
for lib in visionselector_amd/build/variants/libvsel_tr*.so; do
  v=$(basename $lib .so); v=${v#libvsel_}
  echo "$v $(python tools/trace_fwd64.py --variant $v "$@" 2>/dev/null | grep '^{' | tail -1)"
done
