#!/usr/bin/env python3
"""C1 padded-width sweep (S = 24 / 32 / 40 / 56): `bench.py --seq-len S` for every S in one table.

The collator pads a batch to 8 * ceil(max_len / 8) of its longest graph (reference src/data/collator.py:70-111), so the width a
PCQM4M-v2 batch arrives with depends on its longest molecule, not on the typical one; the lengths keep the workload's distribution
(clipped N(22, 6)) at every width.  Writes gpurun_out/<name>.json (copy to profiles/ to keep).

    python tools/seq_sweep.py [--seq 24,32,40,56] [--steps 40] [--warmup 10] [--long-tail 0.0] [--out r06_c1_seq_sweep] [--env K=V ...]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", default="24,32,40,56")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--long-tail", type=float, default=0.0)
    ap.add_argument("--out", default="r06_c1_seq_sweep")
    ap.add_argument("--env", action="append", default=[], help="K=V pairs for the bench processes")
    ap.add_argument("--repeat", type=int, default=1, help="runs per width (the best ms/step is kept, all are listed)")
    a = ap.parse_args()
    env = dict(os.environ)
    for kv in a.env:
        k, v = kv.split("=", 1)
        env[k] = v
    rows = []
    for S in [int(x) for x in a.seq.split(",")]:
        runs = []
        for _ in range(a.repeat):
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup), "--no-cpu-baseline",
                   "--seq-len", str(S)] + (["--long-tail", str(a.long_tail)] if a.long_tail > 0 and S > 32 else [])
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print(r.stdout[-2000:], r.stderr[-4000:], file=sys.stderr)
                raise SystemExit(f"bench.py --seq-len {S} failed")
            runs.append(json.loads(line[-1]))
        j = min(runs, key=lambda x: x["ms_per_step"])
        rows.append({"seq_len": S, "ms_per_step": j["ms_per_step"], "ms_per_step_runs": [x["ms_per_step"] for x in runs],
                     "real_tokens_per_s": j["value"], "padded_tokens_per_s": j["padded_tokens_per_s"],
                     "real_tokens_per_step": j["value"] * j["ms_per_step"] * 1e-3, "rows_per_batch": j["step_mfma"]["rows_per_batch"],
                     "step_mfma_frac_of_peak": j["step_mfma"]["frac_of_peak"], "smtp_loss": j["smtp_loss"],
                     "layouts_ms": (j.get("layouts") or {}).get("ms_per_step"),
                     "per_sample_kernels": {k: j["per_sample_kernels"][k]["avg_launch_ms"] for k in ("forward", "backward")}
                     if j.get("per_sample_kernels") else None})
        print(json.dumps(rows[-1]), flush=True)
    ref = next((r for r in rows if r["seq_len"] == 32), None)
    if ref:
        for r in rows:
            r["real_tokens_per_s_vs_S32"] = r["real_tokens_per_s"] / ref["real_tokens_per_s"]
    out = {"what": "bench.py --seq-len S, workload pcqm4m-v2-pretrain-base (B 256, F 13, V 756, base d768 / L12), lengths clipped N(22, 6) at every width, "
                   "one graph of every batch at the full width; step = fwd + bwd + clip + AdamW, four batches in rotation",
           "steps": a.steps, "warmup": a.warmup, "long_tail": a.long_tail, "env": a.env, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", a.out + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print("S    ms/step   real tok/s   vs S=32")
    for r in rows:
        print(f"{r['seq_len']:<4d} {r['ms_per_step']:.3f}    {r['real_tokens_per_s'] / 1e6:.4f} M   {r.get('real_tokens_per_s_vs_S32', float('nan')):.3f}")


if __name__ == "__main__":
    main()
