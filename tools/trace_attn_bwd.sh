#!/bin/bash
# per-kernel durations of the long-sequence attention backward, dropout 0 / 0.1 and every GGET_ATTN_BWD_VARIANT given
cd /tmp && export TMPDIR=/tmp
for p in 0.0 0.1; do for v in ${@:-0}; do
  d=/tmp/trb_${p}_${v}; rm -rf $d
  P=$p GGET_ATTN_BWD_VARIANT=$v rocprofv3 --kernel-trace --output-format rocpd -d $d -- python $GRAFT_REPO_ROOT/tools/attn_bwd_only.py > $d.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  echo "== p=$p variant=$v"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $db 2>&1 | grep -i attn
done; done
