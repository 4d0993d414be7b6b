"""The C++ facade (include/lvk/LiveVisionKit.hpp) against the reference's call sites: compiles and links on CPU,
runs a synthetic stream on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "plugin_conformance.cpp")


def _build(tmp_path, defines=()):
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    exe = str(tmp_path / "conformance")
    cmd = ["g++", "-std=c++20", "-Wall", "-I" + os.path.join(ROOT, "include"), *defines, "-o", exe, SRC,
           "-L" + os.path.join(ROOT, "livevisionkit_amd"), "-llvk_hip", "-L" + tlib, "-l:libamdhip64.so",
           "-Wl,-rpath," + os.path.join(ROOT, "livevisionkit_amd"), "-Wl,-rpath," + tlib]
    subprocess.check_call(cmd)
    return exe


def test_plugin_call_sites_compile_and_link(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.check_output([exe]).decode()
    assert "conformance TU compiled" in out


@pytest.mark.gpu
def test_plugin_call_sites_run_on_gpu(tmp_path):
    exe = _build(tmp_path, ["-DRUN_ON_GPU"])
    out = subprocess.check_output([exe], timeout=300).decode()
    assert "emitted 7 frames" in out
    assert "scaling ok: Scaling Filter" in out
    assert "composite ok: 6 frames" in out
    assert "chain ok: free-running == synchronous" in out


@pytest.mark.gpu
def test_facade_emits_the_golden_frames(tmp_path):
    """Pixels through the C++ API the plugin links against: lvk::StabilizationFilter::apply on the golden clip, both presets, must give
    the frames of tests/golden/stabilizer.npz (sha-256 per emitted frame; the same vectors pin the oracle and the C-ABI)."""
    import hashlib
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "stabilizer.npz"))
    clip = np.ascontiguousarray(d["clip"])
    n, rows, cols = clip.shape[:3]
    (tmp_path / "clip.raw").write_bytes(clip.tobytes())
    exe = _build(tmp_path, ["-DRUN_ON_GPU"])
    out = subprocess.check_output([exe, "--golden", str(tmp_path / "clip.raw"), str(n), str(rows), str(cols), str(tmp_path / "out.raw")], timeout=300).decode()
    assert f"golden homography: {n - 3} frames" in out and f"golden field: {n - 3} frames" in out
    got = np.frombuffer((tmp_path / "out.raw").read_bytes(), np.uint8).reshape(2, n - 3, rows, cols, 3)
    for k, name in enumerate(("homography", "field")):
        want = d[name + "_sha"][3:]
        for i in range(n - 3):
            sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(got[k, i]).tobytes()).digest(), np.uint8)
            assert np.array_equal(sha, want[i]), (name, i)


@pytest.mark.gpu
def test_facade_throughput_at_4k(tmp_path):
    """Frames/s through lvk::StabilizationFilter::apply at 3840x2160 with resident frames: no per-frame hipMalloc / hipFree (pooled
    frames), so the facade must run at the C-ABI's rate -- the floor here is generous, the figure is printed for DESIGN.md."""
    import re
    exe = _build(tmp_path, ["-DRUN_ON_GPU", "-O2"])
    out = subprocess.check_output([exe, "--bench", "2160", "3840", "600"], timeout=600).decode()
    print(out)
    rates = {m.group(1): float(m.group(2)) for m in re.finditer(r"facade bench 3840x2160 (\S+): (\d+) frames/s", out)}
    assert set(rates) == {"packed", "packed+overlap", "i420+overlap"}
    assert rates["packed"] > 2000 and rates["i420+overlap"] > 3000, rates
