#!/usr/bin/env python3
"""A busy loop on every CPU this process may use, for at most --seconds: the background load of the GPU suite's "noisy host" run
(the driver's `pytest -x -m gpu` must be green on a box whose host cores are busy).  The workers are CHILD processes of
this one (exact PIDs); SIGTERM / SIGINT to the parent ends them.  usage: python tools/cpu_hog.py --seconds 900 &  HOG=$!; ...; kill $HOG"""
import argparse
import multiprocessing as mp
import os
import signal
import sys
import time


def spin(deadline):
    x = 1.0
    while time.time() < deadline:
        for _ in range(200000):
            x = x * 1.0000001 + 1e-9
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--workers", type=int, default=0, help="0 = one per usable CPU")
    a = ap.parse_args()
    n = a.workers or len(os.sched_getaffinity(0))
    deadline = time.time() + a.seconds
    procs = [mp.Process(target=spin, args=(deadline,), daemon=True) for _ in range(n)]
    for p in procs:
        p.start()

    def stop(*_):
        for p in procs:
            if p.is_alive():
                p.terminate()
        sys.exit(0)
    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    print(f"cpu_hog: {n} busy workers for {a.seconds:.0f} s (pid {os.getpid()})", flush=True)
    for p in procs:
        p.join()


if __name__ == "__main__":
    main()
