#!/usr/bin/env python3
"""A/B runs of bench.py under the -DRFE_TUNING build (rover-slam_amd/librover_fe_tuning.so, `make -C rover-slam_amd/csrc tuning`),
one child process per variant (the switches are read once per process).  GPU box only.
usage: python tools/tune_sweep.py 'NAME=ENV1=v,ENV2=v' ...   (NAME alone = baseline);  --repeat N, --steps K"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = ["lg_attention", "conv1ab", "lg_ffn1", "lg_ffn2", "lg_qkv", "lg_cross_qkv", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
          "convPa", "convDa", "convDb", "convPb", "lg_ln_gelu", "lg_assign", "sp_post"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
    a = ap.parse_args()
    rows = []
    for rep in range(a.repeat):
        for v in a.variants:
            name, _, envs = v.partition("=")
            env = dict(os.environ, RFE_LIBRARY=os.path.join(ROOT, "rover-slam_amd", "librover_fe_tuning.so"))
            for kv in filter(None, envs.split(",")):
                k, _, val = kv.partition("=")
                env[k] = val
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "3", "--no-cpu-baseline", "--no-pcie", "--no-pool",
                                "--sustained-steps", "0"] + a.extra.split(), env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:
                print(f"{name}: FAILED rc={r.returncode} {r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else ''}", flush=True)
                continue
            if "stages_ms_per_step" in d:        # latency workloads (--extra '--workload c3'): flat stage table
                st = {k: {"ms_per_step": v} for k, v in d["stages_ms_per_step"].items()}
            else:
                st = d["stages"]
            rows.append((name, d["value"], d["ms_per_step"], {k: st[k]["ms_per_step"] for k in STAGES if k in st}))
            print(f"{name:24s} {d['value']:8.2f} {d['unit']} {d['ms_per_step']:7.4f} ms | " + " ".join(f"{k}={st[k]['ms_per_step']:.4f}" for k in STAGES if k in st), flush=True)


if __name__ == "__main__":
    main()
