#!/bin/bash
# per-kernel durations of the long-sequence attention backward (dQ + dK/dV), dropout 0 and 0.1; extra arguments are passed as
# environment assignments to the run (e.g. GGET_ATTN_BIG=0 for the register-prefetch kernels)
cd /tmp && export TMPDIR=/tmp
for p in 0.0 0.1; do
  d=/tmp/trb_${p}; rm -rf $d
  env P=$p "$@" rocprofv3 --kernel-trace --output-format rocpd -d $d -- python $GRAFT_REPO_ROOT/tools/attn_bwd_only.py > $d.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  echo "== p=$p $*"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $db 2>&1 | grep -i attn
done
