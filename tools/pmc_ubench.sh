#!/bin/bash
# what the SQ counters count: the valu_mfma micro-benchmark's five configurations under the counters used for the attention kernels
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ubench.txt; : > $out
for cfg in 0 1 2 3 4; do
  echo "== config $cfg (0 mfma only, 1 fma only, 2 exp only, 3 all blocked, 4 all interleaved), 4 waves / SIMD, 20000 iterations" >> $out
  for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    rm -rf /tmp/pmu_$c
    rocprofv3 --kernel-trace --pmc $c --output-format rocpd -d /tmp/pmu_$c -- $GRAFT_REPO_ROOT/tools/ubench/valu_mfma $cfg > /tmp/pmu.log 2>&1
    db=$(find /tmp/pmu_$c -name "*.db" | head -1)
    python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $db "k<" >> $out 2>&1
  done
done
cat $out
