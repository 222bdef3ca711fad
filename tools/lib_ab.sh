#!/bin/bash
# Same-box A/B of two builds of the library inside bench.py (alternated processes): tools/lib_ab.sh <base.so> [rounds] [bench args ...]
# prints ms/step of every run; the candidate is the in-tree graph-gpt_amd/lib/libgget_hip.so.
base=$1; rounds=${2:-4}; shift 2
get() { python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), round(j['smtp_loss'],6) if 'smtp_loss' in j else '')"; }
for r in $(seq $rounds); do
  echo "base $(GGET_LIB_PATH=$base python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | get)"
  echo "cand $(python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | get)"
done
