#!/bin/bash
# Whole-step per-kernel MFMA-busy and LDS-conflict table of the default bench run (two separate --pmc passes, kernel-trace only)
# -> gpurun_out/<tag>_pmc_mfma_lds.txt.  usage: pmc_step.sh <tag> [bench.py args]
tag=${1:-r03}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_m /tmp/pmcs_l
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format rocpd -d /tmp/pmcs_m -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /tmp/pmcs_m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format rocpd -d /tmp/pmcs_l -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /tmp/pmcs_l.log 2>&1
dm=$(find /tmp/pmcs_m -name "*.db" | head -1); dl=$(find /tmp/pmcs_l -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_mfma.py $dm $dl > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_mfma_lds.txt 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_mfma_lds.txt
