#!/bin/bash
# tools/dbg/build_asan.sh — HOST-side AddressSanitizer build of the library and of the oracle (device code untouched), for crash hunting with tools/soak.py:
#   tools/dbg/build_asan.sh && cp ab/asan/libcomet_hip.so comet_amd/ && cp ab/asan/libcomet_oracle.so oracle/
#   ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so python tools/soak.py 1000 <seed>
set -e
R=$(cd $(dirname $0)/../.. && pwd); S=$R/comet_amd/csrc; B=$S/build_asan; O=$R/ab/asan; mkdir -p $B $O
FL="--offload-arch=gfx950 -O3 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
cd $S
for f in kernels_dist kernels_select kernels_quant kernels_fast kernels_scanq kernels_scanq_l2 kernels_ivf index_flat index_quant index_text index_hnsw comm api; do
  ( /opt/rocm/bin/hipcc $FL -c $f.hip -o $B/$f.o 2>&1 | grep -i "error" ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan -o $O/libcomet_hip.so $B/*.o -Wl,-rpath,/opt/rocm/lib -ldl
/opt/rocm/lib/llvm/bin/clang++ -O2 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address -shared-libsan -fno-omit-frame-pointer $R/oracle/comet_oracle.cpp -o $O/libcomet_oracle.so
ls -la $O
