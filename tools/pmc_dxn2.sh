#!/bin/bash
# PMC passes (one counter per run, no trace domains besides --kernel-trace) over the dxn2 dgrad launch -> gpurun_out/<tag>_dxn2_pmc.json
# FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of 16 B/lane reads at 64 B).
tag=${1:-r02}
export GGET_T=${2:-5696}     # rows of the launch
cd /tmp && export TMPDIR=/tmp
declare -A val
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format rocpd -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/gemm_dxn2_pmc.py > /tmp/pmc_$c.log 2>&1
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $db gemm_ks_kernel > /tmp/pmc_$c.txt 2>&1
  cat /tmp/pmc_$c.txt
done
python - <<PY
import json, re, hashlib, os
root = os.environ["GRAFT_REPO_ROOT"]
def get(c):
    t = open(f"/tmp/pmc_{c}.txt").read()
    m = re.search(rf"{c}: dispatches (\d+) records/dispatch (\d+) sum/dispatch ([\d.]+) avg_duration_us ([\d.]+)", t)
    return float(m.group(3)), float(m.group(4)), int(m.group(2))
fetch, dur, _ = get("FETCH_SIZE"); write, _, _ = get("WRITE_SIZE"); mfma, _, recs = get("SQ_VALU_MFMA_BUSY_CYCLES"); act, _, arecs = get("GRBM_GUI_ACTIVE")
h = hashlib.sha256()
for f in ("gemm.hip", "common.h", "gemm.h"): h.update(open(os.path.join(root, "graph-gpt_amd", "csrc", f), "rb").read())
T, d, ff = int(os.environ.get("GGET_T", "5696")), 768, 3072
alg = T * 2 * ff * 2 + 2 * ff * d * 2 + T * d * 2
out = {"kernel": f"gemm_ks_kernel<96,192,NN> (in-block K split) on dgu [{T},6144] x W_gu [6144,768] -> dxn2 [{T},768] (dgrad of the gate|up projection, C1)",
       "rows": T,
       "command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/gemm_dxn2_pmc.py (one counter per pass; 5 launches averaged; tools/pmc_dxn2.sh)",
       "source_digest": h.hexdigest()[:16], "avg_duration_us": dur, "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
       "FETCH_SIZE_note": "gfx950 tallies 128-B requests of 16 B/lane reads at 64 B: doubled (MI355X_MICROARCH.md, HBM section)",
       "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "algorithmic_bytes_per_launch": alg,
       "algorithmic_note": "dgu T*6144*2 + W_gu 6144*768*2 read once; dxn2 T*768*2 written once",
       "flops_per_launch": 2.0 * T * 2 * ff * d,
       "SQ_VALU_MFMA_BUSY_CYCLES_sum": mfma, "GRBM_GUI_ACTIVE_sum": act, "mfma_busy_frac": mfma / (1024.0 * dur * 2200.0), "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.2 GHz), as in the whole-step table"}
json.dump(out, open(os.path.join(root, "gpurun_out", "${tag}_dxn2_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
