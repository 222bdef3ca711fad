#!/bin/bash
# kernel trace of the default bench run -> gpurun_out/<tag>_trace.txt (per-kernel table), run on the GPU box
# usage: trace_c1.sh <tag> [extra bench.py args, e.g. --layout padded]
tag=${1:-c1}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db > $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.txt 2>&1
python $GRAFT_REPO_ROOT/tools/gap_stats.py $db > $GRAFT_REPO_ROOT/gpurun_out/${tag}_gaps.txt 2>&1
python $GRAFT_REPO_ROOT/tools/step_seq.py $db > $GRAFT_REPO_ROOT/gpurun_out/${tag}_seq.txt 2>&1
tail -1 /tmp/prof_$tag.log | cut -c1-300
