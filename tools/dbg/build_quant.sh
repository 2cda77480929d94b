#!/bin/bash
# tools/dbg/build_quant.sh [extra flags]: rebuild kernels_quant.o (+ the library) with extra flags (e.g. -DA2_TRACE); stops on the first error
cd /root/repo/comet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function "$@" -c kernels_quant.hip -o build/kernels_quant.o > /tmp/t/build.log 2>&1 || { grep -E "error" /tmp/t/build.log; exit 1; }
make > /tmp/t/make.log 2>&1 || { tail -5 /tmp/t/make.log; exit 1; }
ls -la ../libcomet_hip.so
