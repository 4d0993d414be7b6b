"""Register / scratch budget of every kernel of liblvk_hip.so, read from the gfx950 assembly hipcc emits (no GPU needed).

Round 3 lost 2x on k_mesh_backsolve to an innocent-looking refactor (its LDS array handed to a helper as a generic pointer: 32 VGPRs less,
112 bytes of the walking wavefront's state in scratch, 43 -> 89 us) and only a timeline caught it.  This test catches that class on the
CPU: no kernel may use scratch, and the kernels whose occupancy the schedule depends on stay within their VGPR budgets (DESIGN.md
sections 4-5: the remap's persistent grid needs <= 80 VGPRs to fit 4 blocks per CU next to the tracker; k_ransac_finalize runs 8 waves of
<= 96 VGPRs so that two of them per SIMD fit next to the remap's 4 x 80)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "livevisionkit_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
MESH_FLAGS = ["-mllvm", "-amdgpu-load-store-vectorizer=0", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]      # as csrc/Makefile
VGPR_BUDGET = {r"k_fast_insert": 48, r"k_remap_\w+": 80, r"k_easu_scale": 80, r"k_ransac_finalize": 96, r"k_mesh_backsolve(?!_generic)": 168, r"k_pyrlk": 96, r"k_mesh_solve(?!_generic)": 256}


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
    """The remap is bound by how many VALU instructions it issues per pixel.  Round 4: 498.9 per pixel measured
    (rocprofv3 SQ_INSTS_VALU, profiles/r04_sq_counters_per_kernel.txt) = 2 124 static VALU instructions per 4-pixel thread of
    k_remap_homography_420 (both paths of every branch counted).  Round 5 took the 64-bit address arithmetic out of the per-pixel path: the
    four tap-row loads, the border copy, the mesh / table loads and the sinks address their frames with ONE 32-bit byte offset against
    block-uniform scalar bases (global_load v, v_off, s[base:base+1]) -- 2 066 static VALU for the homography kernel, 2 155 for the mesh
    kernel (was 2 241), no v_mad_u64_u32 anywhere in them and a handful of v_lshl_add_u64 per THREAD (ragged-edge stores) instead of 6-7
    per pixel -- and made `coord - floor(coord)` one v_fract_f32 (exact for every coordinate that reaches the EASU path): 2 058.  This test keeps a refactor from quietly adding to it, and holds the round-4 savings in place: EASU's saturate as the
    `clamp` modifier of the multiply that feeds it and the final clamp between the centre taps' minimum and maximum as one v_med3_f32."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, "remap.hip")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]

    def bodies_of(kernel):
        return {m.group(1): m.group(2) for m in re.finditer(r"^(\S*" + kernel + r"I\S*):[^\n]*\n(.*?)\.Lfunc_end", out.stdout, re.S | re.M)}

    def valu_of(body):
        ops = [ln.split()[0] for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", ".", "//")) and not ln.strip().endswith(":")]
        return [o for o in ops if o.startswith("v_")]
    # (the mesh kernels hold TWO copies of the strip body since round 5 -- the mesh in LDS for meshes up to 2048 values, in global memory beyond --
    #  and run one of them: their static counts are the sum of both)
    for kernel, ceiling, addr64 in (("k_remap_homography_420", 2066, 12), ("k_remap_mesh_420", 2 * 2200, 40), ("k_remap_homography_lens_420", 2245, 12),
                                    ("k_remap_mesh_lens_420", 2 * 2360, 40)):
        bodies = bodies_of(kernel)
        assert len(bodies) == 2, (kernel, list(bodies))                    # <false> (I420) and <true> (NV12)
        for name, body in bodies.items():
            valu = valu_of(body)
            assert len(valu) <= ceiling, f"{name}: {len(valu)} static VALU instructions (ceiling {ceiling})"
            # the tap loads: scalar base + 32-bit offset (no per-pixel 64-bit pointer arithmetic)
            assert not any(o.startswith("v_mad_u64_u32") for o in valu), f"{name}: 64-bit multiply-add in the address path again"
            assert sum(1 for o in valu if o.startswith("v_lshl_add_u64")) <= addr64, f"{name}: 64-bit address adds are back in the per-pixel path"
            taps = [ln for ln in body.splitlines() if re.search(r"global_load_dwordx[23] v\[\d+:\d+\], v\d+, s\[\d+:\d+\]", ln)]
            assert len(taps) >= 16, f"{name}: {len(taps)} tap-row loads in the scalar-base form (16 expected: 4 rows x 4 pixels)"
            clamped = [ln for ln in body.splitlines() if re.search(r"v_mul_f32_e64 .* clamp", ln)]
            assert len(clamped) >= 32, f"{name}: {len(clamped)} clamped multiplies (8 per pixel expected)"
            assert sum(1 for o in valu if o.startswith("v_med3_f32")) >= 12, name
            if "mesh" in kernel:                                   # the LDS copy of the strip body reads its vertices with ds_read
                assert sum(1 for ln in body.splitlines() if ln.strip().startswith("ds_read")) >= 16, f"{name}: the mesh is not read from LDS"
            assert not re.search(r"v_min_f32_e32 v\d+, 1\.0,", body), f"{name}: a saturate compiled to v_min 1.0 / v_max 0 again"
            assert "v_floor_f32" not in body and sum(1 for o in valu if o.startswith("v_fract_f32")) >= 8, f"{name}: the sub-pixel phase is not a v_fract_f32"
