#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the C-ABI library and the C++ drivers ("sanitizers":
# the reference has none).  Device code is NOT instrumented (GPU ASan is not available on this pool): every
# -fsanitize= sits behind -Xarch_host.  Outputs go to ab/san/ (git-ignored); run the drivers there on a GPU box:
#   bash tools/sanitize_host.sh            # build (CPU only)
#   bash tools/sanitize_host.sh run        # on the GPU box: the runs below, each under its own time limit
set -e
cd "$(dirname "$0")/.."
OUT=ab/san
HIPCC=/opt/rocm/bin/hipcc
CLANGXX=/opt/rocm/lib/llvm/bin/clang++
SAN="-fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer"
if [ "${1:-build}" = build ]; then
  mkdir -p $OUT
  $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wall -Wno-unused-function -ldl \
    -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -shared-libsan \
    -Iinclude -Ifluidsimulation_amd/csrc fluidsimulation_amd/csrc/pbf_hip.hip fluidsimulation_amd/csrc/pbf_comm.hip -o $OUT/libpbf_hip.so
  H=fluidsimulation_amd/host
  $CLANGXX -O1 -g -std=c++17 -fPIC $SAN -shared-libsan -Iinclude -I$H -shared -o $OUT/libpbf_host.so $H/Simulation.cpp -L$OUT -lpbf_hip
  $CLANGXX -O1 -g -std=c++17 $SAN -shared-libsan -Iinclude -I$H -o $OUT/pbf_run $H/pbf_run.cpp -L$OUT -lpbf_host -lpbf_hip -Wl,-rpath,'$ORIGIN' \
    -Wl,-rpath,$(dirname $($CLANGXX -print-file-name=libclang_rt.asan-x86_64.so))
  $CLANGXX -O1 -g -std=c++17 $SAN -shared-libsan -pthread -Iinclude -I$H -o $OUT/pbf_slab_run $H/pbf_slab_run.cpp -L$OUT -lpbf_hip -Wl,-rpath,'$ORIGIN' \
    -Wl,-rpath,$(dirname $($CLANGXX -print-file-name=libclang_rt.asan-x86_64.so))
  $CLANGXX -O1 -g -std=c++17 $SAN -shared-libsan -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include -o $OUT/san_driver tools/san_driver.cpp -L$OUT -lpbf_hip \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$(dirname $($CLANGXX -print-file-name=libclang_rt.asan-x86_64.so))
  echo "built $OUT/{libpbf_hip.so,libpbf_host.so,pbf_run,pbf_slab_run,san_driver}"
  exit 0
fi
# run: leaks are not checked (the HIP runtime keeps its own allocations until exit).  quarantine_size_mb: ROCm's ASan
# runtime keeps freed PINNED host memory (hipHostFree) in the quarantine too and aborts (CHECK in
# sanitizer_allocator_device.h) when such a chunk is recycled after the HSA runtime has unloaded, i.e. by the frees of
# the exit handlers once the default 256 MB quarantine is full -- a quarantine that never fills avoids the recycle.
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:quarantine_size_mb=8192 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $OUT
timeout -k 10 200 ./san_driver && \
timeout -k 10 120 ./pbf_run --scene c1 --steps 5 && \
timeout -k 10 120 ./pbf_run --scene c2 --steps 3 && \
timeout -k 10 200 ./pbf_slab_run --dims 96 40 40 --slabs 4 --steps 30 --ghost-layers 2 --check && \
timeout -k 10 200 ./pbf_slab_run --dims 96 40 40 --slabs 4 --steps 30 --ghost-layers 2 --overlap --check && \
timeout -k 10 200 ./pbf_slab_run --dims 96 40 40 --slabs 3 --steps 60 --ghost-layers 4 --rebalance 10 --check && \
echo "sanitized host runs: clean"
