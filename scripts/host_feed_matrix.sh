#!/bin/bash
# variants of the host-fed schedule (lvk_hip_stab_push_yuv420_host), free-running frames/s at 4K; run on the GPU box
run() { echo "== $*"; env "$@" python scripts/host_feed_probe.py 300 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; }
run LOOKAHEAD=1
TAILN=14 run LOOKAHEAD=1 LVK_HIP_HOST_TRACE=1 LVK_HIP_HOST_SINK=copy
run LOOKAHEAD=1 LVK_HIP_HOST_SINK=direct
run LOOKAHEAD=1 LVK_HIP_HOST_SINK=copy LVK_HIP_HOST_D2H=32
run LOOKAHEAD=1 LVK_HIP_HOST_SINK=copy LVK_HIP_HOST_D2H=64
run LOOKAHEAD=1 LVK_HIP_HOST_SINK=copy LVK_HIP_HOST_D2H=256
run LOOKAHEAD=1 LVK_HIP_HOST_SINK=copy LVK_HIP_HOST_D2H=64 LVK_HIP_HOST_H2D=64
run LOOKAHEAD=1 LVK_HIP_HOST_SINK=copy LVK_HIP_HOST_UP2=1
export TMPDIR=/tmp; R=$PWD; cd /tmp
LOOKAHEAD=1 LVK_HIP_HOST_SINK=copy rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/hostfeed4 -- python $R/scripts/host_feed_probe.py 120 > /dev/null 2>&1
