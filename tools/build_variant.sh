#!/bin/bash
# build_variant.sh NAME [-Dflags...]: a copy of libtbk_hip.so whose probe kernels are compiled with the
# given flags, as trio_binning_amd/csrc/variants/NAME.so (selected at run time with TBK_LIBRARY;
# tools/gpu_ab.sh runs same-box A/B comparisons over whatever lies there)
set -e
cd "$(dirname "$0")/../trio_binning_amd/csrc"
name=$1; shift
mkdir -p variants build
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result --offload-arch=gfx950 "$@" -c tbk_kernels.hip -o variants/$name.kernels.o
objs=$(ls build/*.o | grep -v tbk_kernels.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name.so $objs variants/$name.kernels.o -lpthread -lz
rm -f variants/$name.kernels.o
echo built variants/$name.so
