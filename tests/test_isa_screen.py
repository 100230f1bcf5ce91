"""Static race screen of the compiled kernels (round 5): no LDS read may sit between the last matrix instruction of a stage and the `s_waitcnt vmcnt(N)` + `s_barrier`
that orders the NEXT stage's LDS-DMA copies -- the machine scheduler hoists such reads across an `asm volatile(... ::: "memory")` statement unless
`__builtin_amdgcn_sched_barrier(0)` pins it.  conv3x3_t16d_kernel (since round 4, latent) and ffn2_ln_lat_kernel had them; the 16-channel form of the
former produced wrong descriptor rows whenever SuperPoint's two heads ran concurrently (tools/scan_lds_hoist.py; hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_lds_read_hoisted_above_a_dma_barrier():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_lds_hoist.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "clean" in r.stdout, r.stdout + r.stderr
