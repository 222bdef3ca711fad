#!/bin/bash
# PMC passes (one counter per run) over the grouped weight-gradient launch of a decoder layer -> gpurun_out/<tag>_wgrad_pmc.json
# (same recipe as tools/pmc_gu.sh: FETCH_SIZE doubled per MI355X_MICROARCH.md)
tag=${1:-r02}
export GGET_T=${2:-8192}     # rows (= K) of the launch
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmcw_$c
  rocprofv3 --kernel-trace --pmc $c --output-format rocpd -d /tmp/pmcw_$c -- python $GRAFT_REPO_ROOT/tools/gemm_wgrad.py > /tmp/pmcw_$c.log 2>&1
  db=$(find /tmp/pmcw_$c -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $db gemm_ks_kernel > /tmp/pmcw_$c.txt 2>&1
  cat /tmp/pmcw_$c.txt
done
python - <<PY
import json, re, hashlib, os
root = os.environ["GRAFT_REPO_ROOT"]
def get(c):
    t = open(f"/tmp/pmcw_{c}.txt").read()
    m = re.search(rf"{c}: dispatches (\d+) records/dispatch (\d+) sum/dispatch ([\d.]+) avg_duration_us ([\d.]+)", t)
    return float(m.group(3)), float(m.group(4)), int(m.group(2))
fetch, dur, _ = get("FETCH_SIZE"); write, _, _ = get("WRITE_SIZE"); mfma, _, recs = get("SQ_VALU_MFMA_BUSY_CYCLES"); act, _, arecs = get("GRBM_GUI_ACTIVE")
h = hashlib.sha256()
for f in ("gemm.hip", "common.h", "gemm.h"): h.update(open(os.path.join(root, "graph-gpt_amd", "csrc", f), "rb").read())
T, d, ff = int(os.environ.get("GGET_T", "8192")), 768, 3072
# operands read once: dgu [T,2ff], dy [T,d], dqkv [T,3d], xn [T,d], h [T,ff], attn [T,d]; outputs written once: 2ff*d + d*ff + 3d*d + d*d
alg = 2 * (T * (2 * ff + d + 3 * d + d + ff + d) + (2 * ff * d + d * ff + 4 * d * d))
out = {"kernel": f"gemm_ks_kernel<192,192,TN> (in-block K split): grouped weight gradients of one decoder layer, dW = dY^T X for gate|up, down, q|k|v, o; K = T = {T}, 256 tiles = one per CU (C1)",
       "rows": T,
       "command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/gemm_wgrad.py (one counter per pass; 5 launches averaged; tools/pmc_wgrad.sh)",
       "source_digest": h.hexdigest()[:16], "avg_duration_us": dur, "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
       "FETCH_SIZE_note": "gfx950 tallies 128-B requests of 16 B/lane reads at 64 B: doubled (MI355X_MICROARCH.md, HBM section)",
       "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "algorithmic_bytes_per_launch": alg,
       "algorithmic_note": "six activation / gradient matrices of T rows read once, four weight-gradient matrices written once (bf16)",
       "flops_per_launch": 2.0 * T * (2 * ff * d + d * ff + 4 * d * d),
       "SQ_VALU_MFMA_BUSY_CYCLES_sum": mfma, "GRBM_GUI_ACTIVE_sum": act, "mfma_busy_frac": mfma / 1024.0 / (act / max(arecs, 1)) if act else None}
json.dump(out, open(os.path.join(root, "gpurun_out", "${tag}_wgrad_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
