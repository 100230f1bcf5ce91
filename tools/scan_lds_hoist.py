#!/usr/bin/env python3
"""Static screen of the compiled kernels for ONE race pattern: an LDS read that the machine scheduler hoisted above the `s_waitcnt vmcnt(N)` + `s_barrier`
that is supposed to order it behind an LDS-DMA copy (global_load_lds).  Heuristic on the gfx950 ISA: any ds_read between the last matrix instruction of a
stage and the next s_barrier (in a correct kernel the reads of a stage FOLLOW its barrier and precede the matrix instructions that consume them).
Round 5 found conv3x3_t16d_kernel and ffn2_ln_lat_kernel with such reads; `__builtin_amdgcn_sched_barrier(0)` around the wait pins them.
usage: python tools/scan_lds_hoist.py            (compiles every rover-slam_amd/csrc/*.hip to ISA; exit code 1 on a hit)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAR = "--war" in sys.argv     # also flag copy REQUESTS between a stage's last matrix instruction and the next barrier (informational: a request that belongs to the
                              # prologue of a ring, or one whose destination stage every wave left a barrier earlier, is legitimate there)


def scan(asm_text):
    txt = asm_text.split("\n")
    kern, hits = None, {}
    uses_dma = {}
    for i, ln in enumerate(txt):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern = m.group(1)
        if kern and "global_load_lds" in ln:
            uses_dma[kern] = True
        if "s_barrier" in ln and kern:
            j, reads = i - 1, []
            while j > 0 and "v_mfma" not in txt[j] and "s_barrier" not in txt[j] and not re.match(r"^_Z\w+:", txt[j]) and i - j < 40:
                if re.search(r"\bds_read", txt[j]) or (WAR and "global_load_lds" in txt[j]):
                    reads.append(txt[j].strip())
                j -= 1
            if reads and "v_mfma" in txt[j]:
                hits.setdefault(kern, []).append(reads)
    return {k: v for k, v in hits.items() if uses_dma.get(k)}


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for src in sorted(glob.glob(os.path.join(ROOT, "rover-slam_amd", "csrc", "*.hip"))):
            out = os.path.join(d, os.path.basename(src) + ".s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only", "-o", out, src],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(out):
                continue                                   # host-only sources
            for k, v in scan(open(out).read()).items():
                bad += 1
                print(f"{os.path.basename(src)}: {k[:100]}: {len(v)} barrier(s) with an LDS read hoisted above, e.g. {v[0][:2]}")
    print("scan_lds_hoist:", "clean" if not bad else f"{bad} kernel(s) flagged")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
