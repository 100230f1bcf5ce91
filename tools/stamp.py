#!/usr/bin/env python3
"""Identity of the build a profile was taken from.  Run in the build container BEFORE a gpurun profile call:
    python tools/stamp.py            -> writes .build_stamp.json (git HEAD, dirty flag, size + sha256 of librover_fe.so)
tools/profile_round.sh copies it next to its outputs on the GPU box (with the box-side sha256 of the library it really loaded);
tools/refresh_profiles.py prints it at the top of every summary it writes, so a file under profiles/ that predates a kernel change
is visibly stale."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def so_identity(path=None):
    path = path or os.path.join(ROOT, "rover-slam_amd", "librover_fe.so")
    h = hashlib.sha256(open(path, "rb").read()).hexdigest()
    return {"so_size": os.path.getsize(path), "so_sha256": h}


def current():
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "rover-slam_amd", "include", "bench.py"], capture_output=True, text=True).stdout.strip())
    return {"git_head": head, "dirty_sources": dirty, "stamped_utc": time.strftime("%Y-%m-%d %H:%M:%S", time.gmtime()), **so_identity()}


def line(st):
    return (f"Build: git {st.get('git_head', '?')[:12]}{' + uncommitted source changes' if st.get('dirty_sources') else ''}, librover_fe.so "
            f"{st.get('so_size', '?')} B sha256 {st.get('so_sha256', '?')[:16]}"
            + (f" (library loaded on the GPU box: sha256 {st['box_so_sha256'][:16]})" if st.get("box_so_sha256") else "")
            + f", stamped {st.get('stamped_utc', '?')} UTC.")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--box":      # on the GPU box: add what is really there
        st = json.load(open(os.path.join(ROOT, ".build_stamp.json"))) if os.path.exists(os.path.join(ROOT, ".build_stamp.json")) else {}
        st["box_so_sha256"] = so_identity()["so_sha256"]
        json.dump(st, open(sys.argv[2], "w"), indent=1)
    else:
        st = current()
        json.dump(st, open(os.path.join(ROOT, ".build_stamp.json"), "w"), indent=1)
        print(line(st))
