#!/bin/bash
# AddressSanitizer build of the library's HOST code (weight packing, workspace carving, ragged planning, the frontdoor tables ...):
# every .hip source compiled with -fsanitize=address for the host pass only (-fno-gpu-sanitize: device code as shipped), linked into
# tinyvc_amd/libtinyvc_asan.so (git-ignored).  `tools/asan_host.sh build` here (no GPU needed); on the GPU box `tools/asan_host.sh run`
# drives the C ABI through tools/asan_drive.py with the ASan runtime preloaded and writes gpurun_out/asan_host.txt.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/tinyvc_amd/csrc"
# The runtime is gcc's libasan (same __asan_* ABI, version check v8): ROCm's compiler-rt build intercepts hsa_amd_memory_pool_allocate for
# device-side ASan and fails every pool allocation when the device code is not instrumented.  The three __sanitizer_internal_mem* entry
# points newer clang emits are forwarded by a three-line shim.
RT=$(gcc -print-file-name=libasan.so)
if [ "$1" = "build" ]; then
  mkdir -p /tmp/tvc_asan
  FL="--offload-arch=gfx950 -O1 -Xarch_device -O3 -Xarch_device -g0 -g -std=c++17 -fPIC -Wno-unused-function -ffp-contract=on -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
  objs=""
  for f in api ragged frontdoor frontend fft encoder knn knn_general decoder filter_up24s conv48s sola; do      # = tinyvc_amd/build.py SOURCES
    /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/tvc_asan/$f.o &
    objs="$objs /tmp/tvc_asan/$f.o"
  done
  wait
  cat > /tmp/tvc_asan/shim.c <<'EOC'
#include <string.h>
void* __sanitizer_internal_memcpy(void* d, const void* s, size_t n) { return memcpy(d, s, n); }
void* __sanitizer_internal_memmove(void* d, const void* s, size_t n) { return memmove(d, s, n); }
void* __sanitizer_internal_memset(void* d, int c, size_t n) { return memset(d, c, n); }
EOC
  gcc -O1 -fPIC -c /tmp/tvc_asan/shim.c -o /tmp/tvc_asan/shim.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtinyvc_asan.so $objs /tmp/tvc_asan/shim.o      # __asan_* stay undefined: the preloaded runtime provides them
  echo built tinyvc_amd/libtinyvc_asan.so
else
  cd "$ROOT"
  mkdir -p gpurun_out
  TLIB=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
  SLIB=$(gcc -print-file-name=libstdc++.so.6)
  # python itself is not instrumented: leak reports would be the interpreter's; the allocator hooks and the redzones of the library's own heap traffic are what is checked
  # (torch/lib on the library path: dlopen through the interceptor does not see libtorch's RUNPATH; libstdc++ next to the runtime: the runtime resolves __cxa_throw when it starts, before torch would load the C++ library)
  LD_PRELOAD="$RT $SLIB" LD_LIBRARY_PATH=$TLIB:$LD_LIBRARY_PATH ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0 TVC_LIB_PATH=$ROOT/tinyvc_amd/libtinyvc_asan.so \
    python tools/asan_drive.py > gpurun_out/asan_host.txt 2>&1 && echo "asan run: clean" >> gpurun_out/asan_host.txt
  tail -5 gpurun_out/asan_host.txt
fi
