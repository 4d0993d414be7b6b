#!/bin/bash
# Phase times of k_mesh_solve (wall_clock64 inside the kernel).
#   scripts/mesh_timing.sh build <name> [hipcc defines...]   builds livevisionkit_amd/variants/liblvk_hip_mesh_<name>.so with
#                                                            -DLVK_MESH_TIMING=${LVK_MESH_TIMING_LEVEL:-1} and the given defines (where hipcc is)
#   scripts/mesh_timing.sh                                    runs a few solves with every such variant (where a GPU is)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "build" ]; then
  name=$2; shift 2
  V=$R/livevisionkit_amd/variants/liblvk_hip_mesh_$name.so
  T=$(mktemp -d)
  mkdir -p $T/livevisionkit_amd $T/include $R/livevisionkit_amd/variants
  cp -r $R/livevisionkit_amd/csrc $T/livevisionkit_amd/; cp -r $R/include/* $T/include/
  rm -f $T/livevisionkit_amd/csrc/*.o
  sed -i "s/^HIPFLAGS *=/HIPFLAGS = -DLVK_MESH_TIMING=${LVK_MESH_TIMING_LEVEL:-1} $* /" $T/livevisionkit_amd/csrc/Makefile
  if [ -n "${MESH_FLAGS+x}" ]; then make -j8 -C $T/livevisionkit_amd/csrc MESH_FLAGS="$MESH_FLAGS" > /dev/null 2>&1; else make -j8 -C $T/livevisionkit_amd/csrc > /dev/null 2>&1; fi
  cp $T/livevisionkit_amd/liblvk_hip.so $V
  rm -rf $T
  echo built $V
  exit 0
fi
cd $R
for V in $R/livevisionkit_amd/variants/liblvk_hip_mesh_*.so; do
echo "== $(basename $V)"
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
done
