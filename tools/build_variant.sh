#!/bin/bash
# Tuning helper: build a variant of the library with extra -D flags on the fused kernels only.
#   tools/build_variant.sh NAME [-DMK_ABLATE=1 ...]   ->  build_ab/libmink_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
name=$1; shift
for f in mk_generic mk_dense mk_api; do
  if [ ! -f build_ab/$f.o ] || [ mink_amd/csrc/$f.hip -nt build_ab/$f.o ] || [ mink_amd/csrc/mk_device.h -nt build_ab/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c mink_amd/csrc/$f.hip -o build_ab/$f.o &
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c mink_amd/csrc/mk_fused.hip -o build_ab/fused_$name.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared build_ab/fused_$name.o build_ab/mk_generic.o build_ab/mk_dense.o build_ab/mk_api.o -o build_ab/libmink_$name.so
echo build_ab/libmink_$name.so
