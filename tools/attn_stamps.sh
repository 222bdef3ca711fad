#!/bin/bash
# build a measurement copy of the library with s_memtime stamps in the dK/dV attention kernel and print the phase times
# (run on the GPU box).  Output: gpurun_out/attn_stamps.txt
set -e
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/stampbuild
for f in engine gemm kernels; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c graph-gpt_amd/csrc/$f.hip -o /tmp/stampbuild/$f.o & done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DGGET_ATTN_STAMPS -c graph-gpt_amd/csrc/attention.hip -o /tmp/stampbuild/attention.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/stampbuild/libgget_stamps.so /tmp/stampbuild/*.o
GGET_LIB_PATH=/tmp/stampbuild/libgget_stamps.so python tools/attn_stamps.py "$@" > gpurun_out/attn_stamps.txt 2>&1
cat gpurun_out/attn_stamps.txt
