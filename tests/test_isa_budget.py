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


def test_remap_instruction_ceiling():
    """The remap is bound by how many VALU instructions it issues per pixel: 498.9 per pixel measured (rocprofv3
    SQ_INSTS_VALU, profiles/r04_sq_counters_per_kernel.txt) = 2 124 static VALU instructions per 4-pixel thread of k_remap_homography_420
    (both paths of every branch counted).  This keeps a refactor from quietly adding to it, and holds the two round-4 savings in place: EASU's
    saturate as the `clamp` modifier of the multiply that feeds it (no v_min_f32 / v_max_f32 pair with 1.0 / 0 behind it) and the final clamp
    between the centre taps' minimum and maximum as one v_med3_f32."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, "remap.hip")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(\S*k_remap_homography_420\S*):[^\n]*\n(.*?)\.Lfunc_end", out.stdout, re.S | re.M)}
    assert len(bodies) == 2, list(bodies)                                  # <false> (I420) and <true> (NV12)
    for name, body in bodies.items():
        ops = [ln.split()[0] for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", ".", "//")) and not ln.strip().endswith(":")]
        valu = [o for o in ops if o.startswith("v_")]
        assert len(valu) <= 2140, f"{name}: {len(valu)} static VALU instructions (ceiling 2 140; round 3: 2 200)"
        clamped = [ln for ln in body.splitlines() if re.search(r"v_mul_f32_e64 .* clamp", ln)]
        assert len(clamped) >= 32, f"{name}: {len(clamped)} clamped multiplies (8 per pixel expected)"
        assert sum(1 for o in valu if o.startswith("v_med3_f32")) >= 12, name
        assert not re.search(r"v_min_f32_e32 v\d+, 1\.0,", body), f"{name}: a saturate compiled to v_min 1.0 / v_max 0 again"
