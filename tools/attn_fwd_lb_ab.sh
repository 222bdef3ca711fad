#!/bin/bash
# A/B of the long-sequence forward's launch bound (waves per SIMD the 8-wave attn_fwd_dense_kernel is compiled for): the default library
# (4: 128 registers, 22 spilled, two blocks per CU) against builds with -DGGET_FWD_DENSE_MINW=3 / 2 (no spills, one block per CU).
# Build the variants first:  hipcc ... -DGGET_FWD_DENSE_MINW=3 -c csrc/attention.hip -o build/attention_v3.o ; link to lib/libgget_hip_v3.so
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "" _v3 _v2; do
    echo "== lib$v (round $rep)"
    GGET_LIB_PATH=$PWD/graph-gpt_amd/lib/libgget_hip$v.so python tools/attn_bench.py 2>&1 | grep -E "S=(1024|2048).*fwd"
  done
done
