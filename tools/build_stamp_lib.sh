#!/bin/bash
# Developer tool: the -DCH8_STAMP build of the library (in-kernel cycle stamps of bottleneck_chain8_kernel), next to the normal objects:
#   tools/build_stamp_lib.sh && DBMM_LIB=$PWD/tools/_bin/libdbmm_stamp.so python tools/chain8_stamps.py 1024
set -e
cd "$(dirname "$0")/../debiasing-multi-modal_amd"
mkdir -p ../tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../include -DCH8_STAMP -c csrc/bottleneck_chain8.hip -o /tmp/chain8_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/_bin/libdbmm_stamp.so $(ls build/*.o | grep -v chain8) /tmp/chain8_stamp.o
echo built tools/_bin/libdbmm_stamp.so
