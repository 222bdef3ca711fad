#!/bin/bash
# Round-3 record run on the GPU box: default bench line (with the CPU baseline), kernel traces of C1 (both layouts) / C3 / C4, the
# whole-step PMC table, the PMC records of the two heaviest launches at the row count of the bench batch, other workloads.
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_final_bench_default.json 2> gpurun_out/r03_final_bench_default.err
python bench.py --layout padded --no-cpu-baseline > gpurun_out/r03_final_bench_padded.json 2>/dev/null
ROWS=$(python -c "import json; print(json.load(open('gpurun_out/r03_final_bench_default.json'))['step_mfma']['rows'])")
echo rows $ROWS
bash tools/trace_c1.sh r03_final_c1 > /dev/null
bash tools/trace_c1.sh r03_final_c1_padded --layout padded > /dev/null
bash tools/trace_workload.sh ogbl-ppa-finetune-base r03_final_c3 > /dev/null
bash tools/trace_workload.sh longseq-finetune-base r03_final_c4 > /dev/null
bash tools/pmc_step.sh r03_final_c1 > /dev/null
bash tools/pmc_wgrad.sh r03 $ROWS > /dev/null
bash tools/pmc_gu.sh r03 $ROWS > /dev/null
python - <<'PY'
import json, subprocess, sys
out = {}
for w in ("pcqm4m-v2-pretrain-base24", "ogbl-ppa-finetune-base", "longseq-finetune-base", "pcqm4m-v2-pretrain-base-packed"):
    for layout in ("varlen", "padded"):
        r = subprocess.run([sys.executable, "bench.py", "--workload", w, "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--layout", layout],
                           capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1])
            out.setdefault(w, {})[layout] = {k: d[k] for k in ("value", "ms_per_step", "padded_tokens_per_s", "step_mfma", "config") if k in d}
            out[w][layout]["loss"] = d.get("smtp_loss", d.get("task_loss"))
json.dump(out, open("gpurun_out/r03_other_workloads.json", "w"), indent=1)
for w, v in out.items():
    print(w, {l: (round(x["ms_per_step"], 2), round(x["value"])) for l, x in v.items()})
PY
python -c "
import json
d = json.load(open('gpurun_out/r03_final_bench_default.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'smtp_loss')}, d['step_mfma'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
