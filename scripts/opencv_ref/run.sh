#!/bin/bash
# One command for whoever has OpenCV 4.8 (and Eigen 3.4): export the inputs of every third-party call on the hot path, build the dump tool
# against the installed libraries, run it, and hold the oracle (hence, through the GPU suite, the HIP kernels) against what they produce.
# Never built or shipped in this repository's image (no OpenCV, no Eigen, no pkg-config there: SURVEY.md section 8c); tests/test_opencv_hook.py
# keeps the plumbing alive with the oracle standing in for the tool.     usage: bash scripts/opencv_ref/run.sh [work directory]
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
WORK=${1:-/tmp/lvk_cv}
mkdir -p "$WORK"
command -v pkg-config > /dev/null || { echo "pkg-config not found: install OpenCV 4.8 (and Eigen 3.4) with their .pc files"; exit 2; }
pkg-config --exists opencv4 || { echo "pkg-config opencv4 not found (the reference pins OpenCV 4.8.0, Scripts/setup_deb.sh:42)"; exit 2; }
FLAGS="$(pkg-config --cflags --libs opencv4)"
if pkg-config --exists eigen3; then FLAGS="$FLAGS -DLVK_WITH_EIGEN $(pkg-config --cflags eigen3)"; else echo "eigen3 not found: row a10 will be skipped"; fi
python "$HERE/opencv_ref_compare.py" export "$WORK"
g++ -O2 -std=c++17 "$HERE/opencv_ref_dump.cpp" -o "$WORK/opencv_ref_dump" $FLAGS
"$WORK/opencv_ref_dump" "$WORK"
cd "$ROOT" && python "$HERE/opencv_ref_compare.py" compare "$WORK"
