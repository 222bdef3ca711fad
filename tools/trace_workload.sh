#!/bin/bash
# kernel trace of one bench workload -> gpurun_out/<tag>_trace.txt (per-kernel table)
w=${1:-ogbl-ppa-finetune-base}; tag=${2:-wl}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db > $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.txt 2>&1
tail -1 /tmp/prof_$tag.log | cut -c1-200
