#!/bin/bash
# tools/build_variant.sh <name> <unit.hip> <object name in lib/obj> <extra flags...> -- a one-off build of ONE unit with extra
# flags, linked with the other units' current objects into sassy_amd/lib_exp/<name>.so (timing experiments:
# SASSY_HIP_LIBRARY=<that file> makes the Python mirror load it).
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; OBJ=$3; shift 3
mkdir -p sassy_amd/lib_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c sassy_amd/csrc/$UNIT -o sassy_amd/lib_exp/$NAME.o
OBJS=$(ls sassy_amd/lib/obj/*.o | grep -v "/$OBJ$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sassy_amd/lib_exp/$NAME.so $OBJS sassy_amd/lib_exp/$NAME.o
rm sassy_amd/lib_exp/$NAME.o
echo sassy_amd/lib_exp/$NAME.so
