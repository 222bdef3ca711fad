#!/bin/bash
# per-kernel PMC sums for the attention kernels: one counter per pass -> gpurun_out/<tag>_attn_pmc.txt
tag=${1:-r02}; p=${2:-0.0}; which=${3:-fwd}; pat=${4:-attn_}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_attn_pmc.txt
: > $out
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmca_$c
  rocprofv3 --kernel-trace --pmc $c --output-format rocpd -d /tmp/pmca_$c -- python $GRAFT_REPO_ROOT/tools/attn_fwd_only.py $p $which > /tmp/pmca_$c.log 2>&1
  db=$(find /tmp/pmca_$c -name "*.db" | head -1)
  if [ -n "$db" ]; then python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $db $pat >> $out 2>&1; else echo "$c: no db ($(tail -1 /tmp/pmca_$c.log))" >> $out; fi
done
cat $out
