#!/bin/bash
# Builds livevisionkit_amd/variants/liblvk_hip_<name>.so: the library with extra compiler flags (A / B partners of the committed build, loaded
# through LVK_HIP_LIB).  usage: bash scripts/variant_build.sh <name> <flags...>    e.g.  finalize256 -DLVK_FINALIZE_THREADS=256
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d)
mkdir -p $T/livevisionkit_amd $T/include $R/livevisionkit_amd/variants
cp -r $R/livevisionkit_amd/csrc $T/livevisionkit_amd/; cp -r $R/include/* $T/include/
rm -f $T/livevisionkit_amd/csrc/*.o
FLAGS="$*"
case "$FLAGS" in *TIMELINE*|*TIMING*|*TOLERANT*) FLAGS="$FLAGS -DLVK_PROBE_BUILD";; esac      # instrumented kernels are only accepted with this (lvk_hip_internal.hpp)
sed -i "s/^HIPFLAGS *=/HIPFLAGS = $FLAGS /" $T/livevisionkit_amd/csrc/Makefile
make -j8 -C $T/livevisionkit_amd/csrc > /dev/null 2>&1
cp $T/livevisionkit_amd/liblvk_hip.so $R/livevisionkit_amd/variants/liblvk_hip_$NAME.so
rm -rf $T
echo built livevisionkit_amd/variants/liblvk_hip_$NAME.so
