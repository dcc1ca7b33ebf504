#!/bin/bash
# A lab copy of the library with extra compiler flags for ONE translation unit: scripts/lab_build.sh <name> <file.hip> <flags...>
# -> build/lab/lib_<name>.so (objects of the other translation units are taken from cirkit_amd/lib/).
set -e
NAME=$1; SRC=$2; shift 2
mkdir -p build/lab
BASE=$(basename "$SRC" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" -c "$SRC" -o "build/lab/${BASE}_$NAME.o"
OBJS=$(ls cirkit_amd/lib/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/lab/lib_$NAME.so" $OBJS "build/lab/${BASE}_$NAME.o"
echo "build/lab/lib_$NAME.so"
