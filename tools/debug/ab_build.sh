#!/bin/bash
# Build a variant of the library for same-box A/B runs (tools/debug/ab_libs.sh): tools/debug/ab_build.sh <name> <source.hip> [flags]
# compiles ONE kernel source with extra flags (e.g. -DSHMP16_SELF_LATE) and links it with the other objects of the
# current build into tools/debug/_ab/lib<name>.so.
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../../desco_amd/csrc"
make -j8 >/dev/null
OBJ=/tmp/ab_${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=200000 "$@" -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v "^$(basename $SRC .hip).o$")
mkdir -p ../../tools/debug/_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/debug/_ab/lib$NAME.so $OTHERS $OBJ -lgomp
python3 ../../tools/check_isa.py ../../tools/debug/_ab/lib$NAME.so
