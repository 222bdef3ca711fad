#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over the per-sample S <= 32 kernels at the headline shape
# (tools/attn_oproj_bench.py: B 256, S 32, H 12) -> gpurun_out/<tag>_attn_oproj_pmc.json: L2 <-> fabric bytes (FETCH_SIZE doubled per
# MI355X_MICROARCH.md, WRITE_SIZE), L2 hit rate (TCC_HIT_sum / TCC_MISS_sum), MFMA-busy.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmca_$n
  rocprofv3 --kernel-trace --pmc $c --output-format rocpd -d /tmp/pmca_$n -- python $GRAFT_REPO_ROOT/tools/attn_oproj_bench.py > /tmp/pmca_$n.log 2>&1
  db=$(find /tmp/pmca_$n -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $db attn_oproj_fwd_kernel > /tmp/pmca_${n}_fwd.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_kernel.py $db attn_oproj_bwd_kernel > /tmp/pmca_${n}_bwd.txt 2>&1
  cat /tmp/pmca_${n}_fwd.txt /tmp/pmca_${n}_bwd.txt
done
python - <<PY
import json, re, os, glob
root = os.environ["GRAFT_REPO_ROOT"]
out = {"shape": "B 256, S 32, H 12 (d 768), PCQM4M-v2 length distribution: 5669 tokens, var-len rows; attention dropout 0.1",
       "command": "rocprofv3 --kernel-trace --pmc <COUNTERS> -- python tools/attn_oproj_bench.py (one counter group per pass; tools/pmc_attn_oproj.sh)",
       "FETCH_SIZE_note": "gfx950 tallies 128-B requests of 16 B/lane reads at 64 B: doubled (MI355X_MICROARCH.md, HBM section)"}
for which in ("fwd", "bwd"):
    rec = {}
    for path in glob.glob(f"/tmp/pmca_*_{which}.txt"):
        for m in re.finditer(r"(\w+): dispatches (\d+) records/dispatch (\d+) sum/dispatch ([\d.]+) avg_duration_us ([\d.]+)", open(path).read()):
            rec[m.group(1)] = {"sum_per_dispatch": float(m.group(4)), "avg_duration_us": float(m.group(5)), "records": int(m.group(3))}
    d, B = 768, 256
    alg = {"fwd": 2 * (5696 * 3 * d + 3 * 5696 * d + 5696 * d) + B * 12 * 32 * 4 + 5696 * 4,           # qkv read; attn_out, x_mid, xn written; x_in read
           "bwd": 2 * (3 * 5696 * d + 5696 * d + 5696 * 3 * d + 5696 * 3 * d) + 5696 * 4}[which]   # dxn, x_mid, dres read; dx_mid written; qkv read; dqkv written
    r = {"counters": rec, "algorithmic_hbm_bytes_per_launch": alg, "weight_stream_bytes_per_launch": B * d * d * 2,
         "weight_stream_note": "every sample's workgroup reads the whole fragment-major o weight (1.18 MB) through its vector cache: L2 hits, not HBM"}
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        r["traffic_bytes_per_launch"] = int(2 * rec["FETCH_SIZE"]["sum_per_dispatch"] * 1024 + rec["WRITE_SIZE"]["sum_per_dispatch"] * 1024)
    if "TCC_HIT_sum" in rec and "TCC_MISS_sum" in rec:
        hit, miss = rec["TCC_HIT_sum"]["sum_per_dispatch"], rec["TCC_MISS_sum"]["sum_per_dispatch"]
        r["l2_hit_rate"] = hit / max(hit + miss, 1.0)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and "GRBM_GUI_ACTIVE" in rec:
        a = rec["GRBM_GUI_ACTIVE"]
        r["mfma_busy_frac"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"]["sum_per_dispatch"] / 1024.0 / (a["sum_per_dispatch"] / max(a["records"], 1))
    out[which] = r
json.dump(out, open(os.path.join(root, "gpurun_out", "${tag}_attn_oproj_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
