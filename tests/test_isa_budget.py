"""Register / scratch budget of every kernel of liblvk_hip.so, read from the gfx950 assembly hipcc emits (no GPU needed).

Round 3 lost 2x on k_mesh_backsolve to an innocent-looking refactor (its LDS array handed to a helper as a generic pointer: 32 VGPRs less,
112 bytes of the walking wavefront's state in scratch, 43 -> 89 us) and only a timeline caught it.  This test catches that class on the
CPU: no kernel may use scratch, and the kernels whose occupancy the schedule depends on stay within their VGPR budgets (DESIGN.md
sections 4-5: the remap's persistent grid needs <= 80 VGPRs to fit 4 blocks per CU next to the tracker; k_ransac_finalize is compiled for
<= 168 so that it fits next to the remap)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "livevisionkit_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
MESH_FLAGS = ["-mllvm", "-amdgpu-load-store-vectorizer=0", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]      # as csrc/Makefile
VGPR_BUDGET = {r"k_fast_insert": 48, r"k_remap_\w+": 80, r"k_easu_scale": 80, r"k_ransac_finalize": 168, r"k_mesh_backsolve(?!_generic)": 168, r"k_pyrlk": 96, r"k_mesh_solve(?!_generic)": 256}


def _kernels(unit):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *(MESH_FLAGS if unit == "mesh" else []), "-S", "--cuda-device-only", "-o", "-",
                          os.path.join(CSRC, unit + ".hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", out.stdout, re.S):
        body = m.group(2)
        res[m.group(1)] = (int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)),
                           int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)))
    return res


@pytest.mark.parametrize("unit", ["remap", "mesh", "motion", "pyrlk", "fast", "imgproc", "ingest", "sharpen", "draw", "lens"])
def test_no_scratch_and_vgpr_budgets(unit):
    kernels = _kernels(unit)
    assert kernels, unit
    for name, (scratch, vgprs) in kernels.items():
        assert scratch == 0, f"{name}: {scratch} bytes of scratch"
        for pat, budget in VGPR_BUDGET.items():
            if re.search(pat, name):
                assert vgprs <= budget, f"{name}: {vgprs} VGPRs (budget {budget})"
