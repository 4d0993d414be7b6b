#!/bin/bash
LVK_HIP_HOST_TRACE=1 python bench.py --steps 2000 --warmup 50 --pool 64 --no-cpu-baseline --no-pcie --no-configs --no-multi-stream --no-reference-kernel --no-lookahead 2>&1 >/dev/null | grep -A20 "[23][0-9][0-9][0-9] frames" | head -22
