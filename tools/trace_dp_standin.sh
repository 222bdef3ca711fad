#!/bin/bash
# kernel trace of tools/dp_standin.py for ONE configuration (default: round-5 rank menu, 16 stand-in workgroups) -> gpurun_out/dp_standin_seq_<cfg>.txt
cfg=${1:-r5:16}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dps
DP_STANDIN_ONLY=$cfg rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_dps -- python $GRAFT_REPO_ROOT/tools/dp_standin.py > /tmp/prof_dps.log 2>&1
db=$(find /tmp/prof_dps -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/step_seq.py $db > $GRAFT_REPO_ROOT/gpurun_out/dp_standin_seq_${cfg/:/_}.txt 2>&1
tail -3 /tmp/prof_dps.log
