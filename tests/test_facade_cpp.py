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
