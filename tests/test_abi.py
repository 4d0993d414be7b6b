"""CPU checks of the drop-in boundary: liblvk_hip.so builds, loads, and exports every symbol include/lvk_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lvk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lvk_(?:hip|stab)_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_python_binds():
    from livevisionkit_amd import _native
    assert _declared_symbols() == _native.symbols()


def test_library_exports_every_declared_symbol():
    from livevisionkit_amd import _native
    lib = _native.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"liblvk_hip.so does not export {name}"


def test_abi_version_of_header_and_library_agree():
    """include/lvk_hip.h groups its declarations into PART 1 (stable ABI) and PART 2 (experimental / diagnostics); the number of PART 1 is
    compiled into the library.  A host built against another PART 1 can tell before it calls anything."""
    from livevisionkit_amd import _native
    text = open(os.path.join(ROOT, "include", "lvk_hip.h")).read()
    want = int(re.search(r"#define LVK_HIP_ABI_VERSION (\d+)", text).group(1))
    lib = _native.load()
    assert lib.lvk_hip_abi_version() == want
    assert ("ABI %d" % want).encode() in lib.lvk_hip_version()
    assert "PART 1 -- STABLE ABI" in text and "PART 2 -- EXPERIMENTAL / DIAGNOSTICS" in text
    # ABI 6 (round 6): every push flavour takes the capacity of its output and reports the geometry it wrote (the delayed frame's own size)
    assert want >= 6
    for decl in (r"lvk_hip_stab_push\(.*?int out_step, int out_rows,.*?lvk_frame_info\* emitted\)", r"lvk_hip_stab_push_yuv420\(.*?int ov_step, int o_rows,.*?lvk_frame_info\* emitted\)",
                 r"lvk_hip_stab_push_yuv420_host\(.*?int ov_step, int o_rows,.*?lvk_frame_info\* emitted\)"):
        assert re.search(decl, text, re.S), decl
    stable, experimental = text.split("PART 2 -- EXPERIMENTAL / DIAGNOSTICS  (no ABI promise")
    # what a host of the reference binds sits in PART 1 ...
    for name in ("lvk_hip_stab_push(", "lvk_hip_stab_push_yuv420(", "lvk_hip_stab_push_yuv420_host(", "lvk_hip_stab_configure(", "lvk_hip_ctx_create(", "lvk_hip_malloc(",
                 "lvk_hip_remap_homography(", "lvk_hip_upscale(", "lvk_hip_stab_set_overlap(", "lvk_hip_device_count(", "lvk_hip_device_usable(",
                 "lvk_hip_stab_next_output("):
        assert name in stable and name not in experimental, name
    # ... the per-stage test entry points, taps and profiling in PART 2
    for name in ("lvk_hip_fast_detect(", "lvk_hip_pyrlk(", "lvk_hip_estimate_global_motion(", "lvk_hip_mesh_solver_solve(", "lvk_hip_stab_get_stats(",
                 "lvk_hip_stab_set_profiling(", "lvk_hip_stab_prefetch_yuv420(", "lvk_hip_native_rcp(", "lvk_hip_stab_schedule_counters("):
        assert name in experimental and name not in stable, name


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/lvk_hip.h compiles as C99 with -pedantic (no C++ in the signatures, no torch / OpenCV types), and a C
    translation unit that uses its types links against the library."""
    import subprocess
    src = tmp_path / "c_abi.c"
    src.write_text('#include "lvk_hip.h"\n'
                   'int main(void) { lvk_frame_info i = {0, 0, 0}; lvk_stab_settings s; lvk_stab_default_settings(&s); (void)i;\n'
                   '  return (lvk_hip_abi_version() == LVK_HIP_ABI_VERSION && s.predictive_samples == 10) ? 0 : 1; }\n')
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    exe = str(tmp_path / "c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L" + os.path.join(ROOT, "livevisionkit_amd"), "-llvk_hip", "-L" + tlib, "-l:libamdhip64.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "livevisionkit_amd"), "-Wl,-rpath," + tlib])
    assert subprocess.run([exe]).returncode == 0


def test_environment_knobs_are_documented_with_their_tests():
    """Every getenv of the product is listed in INTEGRATION.md section 4 together with the test that exercises it (round-4 VERDICT: every rejected
    experiment that keeps a code path is a path no test pins)."""
    knobs = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "livevisionkit_amd")):
        if "variants" in dirpath:
            continue
        for f in files:
            if f.endswith((".hip", ".hpp", ".py")):
                knobs |= set(re.findall(r'getenv\("(LVK_[A-Z0-9_]+)"\)|environ(?:\.get)?\(?\[?"(LVK_[A-Z0-9_]+)"', open(os.path.join(dirpath, f), errors="ignore").read()))
    knobs = {a or b for a, b in knobs}
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = doc[doc.index("## 4. Diagnostics (environment)"):]
    rows = {m.group(1): m.group(0) for m in re.finditer(r"^\| `(LVK_[A-Z0-9_]+)[^|]*\|.*$", section, re.M)}
    assert knobs and knobs <= set(rows), (sorted(knobs), sorted(rows))
    for k in knobs:
        assert re.search(r"tests/\w+\.py|scripts/\w+\.sh", rows[k]), f"{k}: no test named in INTEGRATION.md section 4"
        for t in re.findall(r"(tests/\w+\.py)", rows[k]):
            assert os.path.exists(os.path.join(ROOT, t)), t


def test_no_source_file_of_the_library_is_a_monolith():
    """csrc/ stays navigable: no translation unit above 1 200 lines (round-4 VERDICT; stabilizer.hip was 1 917, mesh.hip 1 351)."""
    csrc = os.path.join(ROOT, "livevisionkit_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            n = sum(1 for _ in open(os.path.join(csrc, f), errors="ignore"))
            assert n <= 1200, f"{f}: {n} lines"


def test_no_device_fails_loudly_not_silently():
    """Without a GPU the product path must refuse to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from livevisionkit_amd import _native
    import livevisionkit_amd as lvk
    lib = _native.load()
    handle = ctypes.c_void_p()
    rc = lib.lvk_hip_ctx_create(0, ctypes.byref(handle))
    assert rc != 0 and handle.value is None
    assert b"device" in lib.lvk_hip_last_error(None).lower()
    with pytest.raises(Exception):
        lvk.Context(0)


def test_product_never_touches_the_oracle():
    """Nothing under livevisionkit_amd/ or include/ may reference oracle/ (it is test infrastructure)."""
    bad = []
    for base in ("livevisionkit_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"lvko_|liblvk_oracle|oracle/|oracle_lib", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps each C-ABI symbol to the reference interface it replaces."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "lvk_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set(re.findall(r"\b(lvk_(?:hip|stab)_[a-z0-9_]+)\s*\(", header))
    assert not [n for n in sorted(names) if n not in doc]
