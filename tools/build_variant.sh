#!/bin/bash
# Build a variant of libvsel.so for same-box A/B runs (tools/ab_attn.sh): recompile ONE source with extra flags, link it with
# the other objects of the regular build.   tools/build_variant.sh NAME SOURCE.hip [-DFLAG ...]  ->  tools/variants/libvsel_NAME.so
set -e
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/visionselector_amd/build
mkdir -p $ROOT/tools/variants /tmp/vsel_variants
python -m visionselector_amd.build >/dev/null
OBJ=/tmp/vsel_variants/${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c $ROOT/visionselector_amd/csrc/$SRC -o $OBJ 2> /tmp/vsel_variants/${NAME}.remarks || { cat /tmp/vsel_variants/${NAME}.remarks | grep -v remark | head -30; exit 1; }
grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize" /tmp/vsel_variants/${NAME}.remarks | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - | sed 's/_ZN4vsel[0-9a-z]*//; s/EPK[A-Za-z0-9_]*//; s/EvPK[A-Za-z0-9_]*//'
OTHERS=$(ls $B/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/variants/libvsel_${NAME}.so $OBJ $OTHERS
echo "built tools/variants/libvsel_${NAME}.so"
