#!/bin/bash
# Tuning helper: build a variant of the library with extra -D flags on the fused kernels only.
#   tools/build_variant.sh NAME [-DMK_ABLATE=1 ...]   ->  build_ab/libmink_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
name=$1; shift
# flags that name MK_DENSE_* go to the wave-per-problem kernels instead of the fused ones
dflags=(); fflags=()
for a in "$@"; do case "$a" in *MK_DENSE_*) dflags+=("$a");; *) fflags+=("$a");; esac; done
if [ ${#dflags[@]} -gt 0 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "${dflags[@]}" -c mink_amd/csrc/mk_dense.hip -o build_ab/dense_$name.o
fi
for f in mk_generic mk_dense mk_lie mk_api; do
  if [ ! -f build_ab/$f.o ] || [ mink_amd/csrc/$f.hip -nt build_ab/$f.o ] || [ mink_amd/csrc/mk_device.h -nt build_ab/$f.o ] || [ mink_amd/csrc/mk_generic.h -nt build_ab/$f.o ] || [ include/mink_amd.h -nt build_ab/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c mink_amd/csrc/$f.hip -o build_ab/$f.o &
  fi
done
if [ ${#fflags[@]} -gt 0 ] || [ ! -f build/obj/mk_fused.hip.o ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-sched-strategy=max-ilp "${fflags[@]}" -c mink_amd/csrc/mk_fused.hip -o build_ab/fused_$name.o
else
  cp build/obj/mk_fused.hip.o build_ab/fused_$name.o
fi
wait
dobj=build_ab/mk_dense.o; [ ${#dflags[@]} -gt 0 ] && dobj=build_ab/dense_$name.o
# (the cooperative and float32 kernels come from the regular build: python -c 'import __graft_entry__ as g; g.build()')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared build_ab/fused_$name.o build_ab/mk_generic.o $dobj build_ab/mk_lie.o build_ab/mk_api.o \
  build/obj/mk_coop.hip.o build/obj/mk_fused32.hip.o -o build_ab/libmink_$name.so
echo build_ab/libmink_$name.so
