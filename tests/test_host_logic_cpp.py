"""The product's host logic (livevisionkit_amd/csrc/host_logic.hpp) on the CPU against KNOWN ANSWERS derived by hand from the cited
reference lines, independent of oracle/: FeatureDetector::detect / propagate priority rules (FeatureDetector.cpp:138-157,182-205),
PathSmoother::next against a closed form of its Gaussian recurrence (PathSmoother.cpp:84-135), WarpMesh arithmetic (WarpMesh.cpp:333-417),
the static mesh constraints (FrameTracker.cpp:380-457).  tests/cpp/host_logic_test.cpp, g++ only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_against_oracle_and_known_answers(tmp_path, oracle):
    exe = str(tmp_path / "host_logic_test")
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "livevisionkit_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
           "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_logic_test.cpp"),
           "-L" + os.path.join(ROOT, "oracle"), "-llvk_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-pthread"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host logic ok" in out.stdout, out.stdout + out.stderr
