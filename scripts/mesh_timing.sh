#!/bin/bash
# Phase times of k_mesh_solve (wall_clock64 inside the kernel): builds livevisionkit_amd/variants/liblvk_hip_meshtiming.so
# (-DLVK_MESH_TIMING) where hipcc is available, and runs a few solves with it where a GPU is.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/livevisionkit_amd/variants/liblvk_hip_meshtiming.so
if [ "$1" = "build" ]; then
  T=$(mktemp -d)
  mkdir -p $T/livevisionkit_amd $T/include $R/livevisionkit_amd/variants
  cp -r $R/livevisionkit_amd/csrc $T/livevisionkit_amd/; cp -r $R/include/* $T/include/
  rm -f $T/livevisionkit_amd/csrc/*.o
  sed -i "s/^HIPFLAGS *=/HIPFLAGS = -DLVK_MESH_TIMING=${LVK_MESH_TIMING_LEVEL:-1} ${LVK_MESH_EXTRA} /" $T/livevisionkit_amd/csrc/Makefile
  make -j8 -C $T/livevisionkit_amd/csrc > /dev/null
  cp $T/livevisionkit_amd/liblvk_hip.so $V
  rm -rf $T
  echo built $V
  exit 0
fi
cd $R
LVK_HIP_LIB=$V python - <<'PY'
import numpy as np, time
import livevisionkit_amd as lvk
ctx = lvk.Context(0)
dev = ctx.mesh_solver(16, 16, gen_region=(480, 270), max_points=2048)
rng = np.random.default_rng(0)
a = np.c_[rng.uniform(2, 440, 900), rng.uniform(2, 240, 900)].astype(np.float32); b = a + 1.0
for i in range(3):
    t = time.perf_counter(); rc = dev.solve(a, b)[0]; print("solve rc", rc, "host time %.1f us" % ((time.perf_counter() - t) * 1e6))
PY
