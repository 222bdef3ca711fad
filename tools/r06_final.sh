#!/bin/bash
# Round-5 record run on the GPU box.  Order matters: the PMC records of the two heaviest launches are taken FIRST and copied into
# profiles/ on the box, so that the default bench line that follows finds the `roofline.traffic` record of the sources it runs.
cd $GRAFT_REPO_ROOT
ROWS=${1:-5696}
bash tools/pmc_wgrad.sh r06 $ROWS > /dev/null && cp gpurun_out/r06_wgrad_pmc.json profiles/r06_wgrad_gemm_pmc_T$ROWS.json
bash tools/pmc_gu.sh r06 $ROWS > /dev/null && cp gpurun_out/r06_gu_pmc.json profiles/r06_gu_geglu_gemm_pmc_T$ROWS.json
bash tools/pmc_dh.sh r06 $ROWS > /dev/null
bash tools/pmc_dxn2.sh r06 $ROWS > /dev/null && cp gpurun_out/r06_dxn2_pmc.json profiles/r06_dxn2_gemm_pmc_T$ROWS.json
bash tools/pmc_attn_oproj.sh r06 > /dev/null
python tools/attn_oproj_bench.py > gpurun_out/r06_attn_oproj_bench.txt 2>&1
S=40 python tools/attn_oproj_bench.py > gpurun_out/r06_attn_oproj_bench_S40.txt 2>&1
S=56 python tools/attn_oproj_bench.py > gpurun_out/r06_attn_oproj_bench_S56.txt 2>&1
python bench.py > gpurun_out/r06_final_bench_default.json 2> gpurun_out/r06_final_bench_default.err
bash tools/trace_c1.sh r06_final_c1 > /dev/null
bash tools/trace_c1.sh r06_final_c1_S40 --seq-len 40 --layout varlen-count > /dev/null
bash tools/pmc_step.sh r06_final_c1 > /dev/null
bash tools/trace_workload.sh ogbl-ppa-finetune-base r06_final_c3 > /dev/null
bash tools/trace_workload.sh longseq-finetune-base r06_final_c4 > /dev/null
python - <<'PY'
import json, subprocess, sys
out = {}
for w in ("pcqm4m-v2-pretrain-base24", "ogbl-ppa-finetune-base", "longseq-finetune-base", "pcqm4m-v2-pretrain-base-packed"):
    r = subprocess.run([sys.executable, "bench.py", "--workload", w, "--steps", "8", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        d = json.loads(line[-1])
        out[w] = {k: d[k] for k in ("value", "ms_per_step", "padded_tokens_per_s", "step_mfma", "config", "layouts") if k in d}
        out[w]["loss"] = d.get("smtp_loss", d.get("task_loss"))
json.dump(out, open("gpurun_out/r06_other_workloads.json", "w"), indent=1)
for w, v in out.items():
    print(w, round(v["ms_per_step"], 2), round(v["value"]), (v.get("layouts") or {}).get("ms_per_step"))
PY
python -c "
import json
d = json.load(open('gpurun_out/r06_final_bench_default.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'smtp_loss')}, d['step_mfma']['frac_of_peak'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['loss_parity']['rel'], d['layouts']['ms_per_step'])"
