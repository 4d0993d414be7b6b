"""SURVEY.md section 5's sanitizer build option: the CPU oracle (every stage, tests/cpp/oracle_sanitize.cpp) and the product's host
logic (tests/cpp/host_logic_test.cpp: livevisionkit_amd/csrc/host_logic.hpp against known answers and the oracle) compiled with
-DLVK_SANITIZE -fsanitize=address,undefined -fno-sanitize-recover=all and run.  `make -C oracle SANITIZE=1` builds liblvk_oracle.so
with the same flags (for LD_PRELOAD=libasan runs of the Python suite)."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-DLVK_SANITIZE", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
COMMON = ["g++", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mavx2", "-mfma", "-Wall", "-Wno-unused-function", "-pthread"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")


def _oracle_sources():
    return sorted(glob.glob(os.path.join(ROOT, "oracle", "*.cpp")))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(COMMON + SAN + ["-I" + os.path.join(ROOT, "oracle"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.cpp")] + _oracle_sources())
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0 and "oracle sanitize ok" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]


def test_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_logic_san")
    subprocess.check_call(COMMON + SAN + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "livevisionkit_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
                                          "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_logic_test.cpp")] + _oracle_sources())
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0 and "host logic ok" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]
