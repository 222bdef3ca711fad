#!/usr/bin/env python3
"""C1 step against the batches' var-len ROW COUNT: `bench.py --mean-len m` for a range of mean graph lengths (B 256, S 32 or --seq-len).
Every GEMM launch plan is chosen from the row count T of the batch; real epochs deliver a different T every step, so a plan that is
tuned at the headline's T = 5 696 and falls off a cliff a few dozen rows away shows up here as a dip in real tokens/s.
Writes gpurun_out/<out>.json.

    python tools/rows_sweep.py [--means 17,18,...,27] [--seq-len 32] [--steps 30] [--out r06_c1_rows_sweep] [--env K=V ...]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--means", default="16,17,18,19,20,21,22,23,24,25,26")
    ap.add_argument("--seq-len", type=int, default=32)
    ap.add_argument("--batches", default="", help="comma list of per-GPU batch sizes: sweep B at the workload's mean length instead of the mean length")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--out", default="r06_c1_rows_sweep")
    ap.add_argument("--env", action="append", default=[])
    a = ap.parse_args()
    env = dict(os.environ)
    for kv in a.env:
        k, v = kv.split("=", 1)
        env[k] = v
    rows = []
    points = [("mean", float(x)) for x in a.means.split(",")] if not a.batches else [("batch", int(x)) for x in a.batches.split(",")]
    for kind_, m in points:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup), "--no-cpu-baseline",
               "--seq-len", str(a.seq_len), "--layout", "varlen-count"] + (["--mean-len", str(m)] if kind_ == "mean" else ["--per-gpu-batch", str(m)])
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            print(r.stdout[-2000:], r.stderr[-4000:], file=sys.stderr)
            raise SystemExit(f"bench.py --mean-len {m} failed")
        j = json.loads(line[-1])
        rows.append({"mean_len": m if kind_ == "mean" else 22.0, "per_gpu_batch": j["config"]["per_gpu_batch"], "rows_per_batch": j["step_mfma"]["rows_per_batch"], "ms_per_step": j["ms_per_step"],
                     "real_tokens_per_s": j["value"], "us_per_1k_rows": j["ms_per_step"] * 1e3 / (sum(j["step_mfma"]["rows_per_batch"]) / len(j["step_mfma"]["rows_per_batch"]) / 1e3),
                     "step_mfma_frac_of_peak": j["step_mfma"]["frac_of_peak"]})
        print(json.dumps(rows[-1]), flush=True)
    out = {"what": "bench.py --mean-len m --layout varlen-count, workload pcqm4m-v2-pretrain-base (B 256), four batches in rotation", "seq_len": a.seq_len,
           "steps": a.steps, "env": a.env, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", a.out + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print("mean  rows/batch (rotation)            ms/step   real tok/s   us per 1k rows")
    for r in rows:
        print(f"{r['mean_len']:<5.1f} B={r['per_gpu_batch']:<4d} {str(r['rows_per_batch']):34s} {r['ms_per_step']:.3f}    {r['real_tokens_per_s'] / 1e6:.4f} M   {r['us_per_1k_rows']:.1f}")


if __name__ == "__main__":
    main()
