"""The compiler's resource remarks for every kernel of the library (tools/kernel_resources.py; cross-compiles, no GPU): no kernel may spill
vector registers or use scratch memory, and the aggregation kernels keep the register counts their workgroups-per-CU figures rest on
(512 registers per SIMD lane, 512-thread workgroups = two waves per SIMD and workgroup)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("c++filt")), reason="needs hipcc + c++filt")
def test_no_kernel_spills_and_the_aggregation_forms_keep_their_register_budgets():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines()[1:]:
        if line.startswith("#") or not line.strip():
            continue
        *name, vgpr, agpr, sgpr, vspill, sspill, scratch, lds, waves = line.split()
        rows[" ".join(name)] = dict(vgpr=int(vgpr), agpr=int(agpr), vspill=int(vspill), scratch=int(scratch), lds=int(lds), waves=int(waves))
    assert len(rows) > 90 and all(r["vspill"] == 0 and r["scratch"] == 0 for r in rows.values())
    # plain form, one or two row blocks per wave: 80 registers and 52 KiB = three workgroups per CU (6 waves per SIMD)
    for rb in (1, 2):
        r = rows["k_aggregate_mfma<%d, false>" % rb]
        assert r["vgpr"] <= 80 and r["waves"] >= 6 and 3 * r["lds"] <= 160 * 1024, r
    # the forms that make layer 1 and the four-row-block forms: two workgroups per CU (128 registers at most)
    for key in ("k_aggregate_mfma<1, true>", "k_aggregate_mfma<2, true>", "k_aggregate_mfma<4, true>", "k_aggregate_mfma<4, false>"):
        r = rows[key]
        assert r["vgpr"] <= 128 and r["waves"] >= 4 and 2 * r["lds"] <= 160 * 1024, (key, r)
    # the H.W kernel: one workgroup of 512 threads per CU (two waves per SIMD), all of it in architectural registers
    for epi in (0, 1):
        r = rows["k_gemm_bf16x6<%d>" % epi]
        assert r["vgpr"] <= 256 and r["agpr"] == 0 and r["waves"] >= 2, r
