#!/bin/bash
# Builds livevisionkit_amd/variants/liblvk_hip_timeline.so: the library with -DLVK_TIMELINE (in-kernel start / end stamps per kernel).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/livevisionkit_amd $T/include $R/livevisionkit_amd/variants
cp -r $R/livevisionkit_amd/csrc $T/livevisionkit_amd/; cp -r $R/include/* $T/include/
rm -f $T/livevisionkit_amd/csrc/*.o
sed -i 's/^HIPFLAGS *=/HIPFLAGS = -DLVK_TIMELINE /' $T/livevisionkit_amd/csrc/Makefile
make -C $T/livevisionkit_amd/csrc > /dev/null
cp $T/livevisionkit_amd/liblvk_hip.so $R/livevisionkit_amd/variants/liblvk_hip_timeline.so
rm -rf $T
echo built livevisionkit_amd/variants/liblvk_hip_timeline.so
