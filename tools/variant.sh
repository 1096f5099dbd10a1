#!/bin/bash
# Build a variant of libdsp.so with extra compiler defines for pipe_kernel.hip (A/B experiments):
#   tools/variant.sh NAME -DEXP_FOO ...   ->  tools/_build/libdsp_NAME.so   (use with DSP_AMD_LIBRARY=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_build
src=${VARIANT_SRC:-pipe_kernel}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
    -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Idsp_amd/csrc "$@" -c dsp_amd/csrc/$src.hip -o tools/_build/${src}_$name.o
objs=$(ls dsp_amd/csrc/build/*.o | grep -v "/$src.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/_build/libdsp_$name.so $objs tools/_build/${src}_$name.o \
    $(gcc -print-libgcc-file-name) -lm
echo built tools/_build/libdsp_$name.so
