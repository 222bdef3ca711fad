#!/usr/bin/env python3
"""The first run on a box with >= 2 GPUs: `bench.py --gpus N` for every N in {2, 4, 8} the box has, over
    GGET_DP_RESERVE_CUS 0 / 32  x  GGET_DP_OVERLAP 1 / 0  x  GGET_DP_BACKEND torch / abi
(explicit settings: the in-bench start-up probe stays out of the way), plus one run per N with everything left to that probe, and N = 1.
One table: ms/step, whole-job tokens/s, speed-up over N = 1, exposed_comm_ms, replicas_bit_identical, distinct devices.
Writes gpurun_out/scale_first_run.json (copy to profiles/).  No multi-GPU number of this engine exists before this script has run
(DESIGN.md section 6); nothing here is a model.

    python tools/scale_first_run.py [--steps 20] [--warmup 5] [--workload pcqm4m-v2-pretrain-base] [--gpus 2,4,8] [--quick]
reference: src/utils/opt_utils.py:13 (DDP all-reduce), src/utils/misc_utils.py:519-526 (one process per GPU).
"""
import argparse
import itertools
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(n, env_extra, a):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = ["--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup), "--workload", a.workload, "--no-cpu-baseline"]
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + tail
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"error": (r.stderr or r.stdout)[-1500:]}
    return json.loads(line[-1])


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="pcqm4m-v2-pretrain-base")
    ap.add_argument("--gpus", default="2,4,8")
    ap.add_argument("--quick", action="store_true", help="only the probe-driven run per N")
    a = ap.parse_args()
    have = torch.cuda.device_count()
    ns = [n for n in (int(x) for x in a.gpus.split(",")) if n <= have]
    if not ns:
        print(f"this box has {have} GPU(s): nothing to scale over (the script needs >= 2)")
        return 0
    rows = []
    base = run(1, {}, a)
    rows.append({"n_gpus": 1, "menu": "single GPU", "ms_per_step": base.get("ms_per_step"), "tokens_per_s": base.get("value"), "error": base.get("error")})
    v1 = base.get("value")
    for n in ns:
        grid = [] if a.quick else list(itertools.product(("0", "32"), ("1", "0"), ("torch", "abi")))
        runs = [({}, "start-up probe decides")] + [({"GGET_DP_RESERVE_CUS": r, "GGET_DP_OVERLAP": o, "GGET_DP_BACKEND": b},
                                                    f"reserve {r} / overlap {o} / {b}") for r, o, b in grid]
        for env_extra, name in runs:
            d = run(n, env_extra, a)
            dp = d.get("dp") or {}
            rows.append({"n_gpus": n, "menu": name, "ms_per_step": d.get("ms_per_step"), "tokens_per_s": d.get("value"),
                         "speedup_over_1": (d["value"] / v1) if d.get("value") and v1 else None,
                         "exposed_comm_ms": dp.get("exposed_comm_ms"), "replicas_bit_identical": dp.get("replicas_bit_identical"),
                         "distinct_devices": dp.get("distinct_devices"), "backend": dp.get("backend"), "menu_probe": dp.get("menu_probe"),
                         "rank_devices": dp.get("rank_devices"), "error": d.get("error")})
            r_ = rows[-1]
            print(f"N={n} {name:34s} " + (f"ERROR {r_['error'][-200:]}" if r_["error"] else
                  f"{r_['ms_per_step']:.3f} ms/step  {r_['tokens_per_s'] / 1e6:.3f} M tok/s  x{r_['speedup_over_1'] or 0:.2f}  exposed {r_['exposed_comm_ms']:.3f} ms  "
                  f"identical {r_['replicas_bit_identical']}  devices {r_['distinct_devices']}"), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "scale_first_run.json"), "w") as f:
        json.dump({"workload": a.workload, "steps": a.steps, "gpus_on_box": have, "rows": rows}, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
