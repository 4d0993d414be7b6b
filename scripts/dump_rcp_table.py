"""Dumps the device's v_rcp_f32 over the 2^23 binary32 mantissas of [1, 2) as deltas (in ulps) to the correctly rounded
reciprocal: tests/golden/gfx950_rcp.npz.  This is DEVICE DATA (what `native_recip` of the reference's OpenCL kernels evaluates to
on gfx950), read through lvk_hip_native_rcp; the oracle's native_rcp() model is built from it (oracle/easu.cpp).
Usage (GPU box): python scripts/dump_rcp_table.py [out.npz]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import livevisionkit_amd as lvk  # noqa: E402


def device_table(ctx):
    bits = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000))
    x = torch.from_numpy(bits.view(np.float32)).cuda()
    r = ctx.native_rcp(x)
    ctx.sync()
    return bits.view(np.float32), r.cpu().numpy()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gfx950_rcp.npz")
    ctx = lvk.Context(0)
    x, r = device_table(ctx)
    cr = (1.0 / x.astype(np.float64)).astype(np.float32)
    delta = (r.view(np.uint32).astype(np.int64) - cr.view(np.uint32).astype(np.int64))
    assert np.abs(delta).max() <= 1
    # 2 bits per mantissa: 0 = exact, 1 = +1 ulp, 3 = -1 ulp
    code = (delta & 3).astype(np.uint8)
    packed = (code[0::4] | (code[1::4] << 2) | (code[2::4] << 4) | (code[3::4] << 6)).astype(np.uint8)
    np.savez_compressed(out, packed=packed, device=np.array(torch.cuda.get_device_name(0)))
    print("wrote", out, os.path.getsize(out), "bytes; deltas -1/0/+1:", int((delta == -1).sum()), int((delta == 0).sum()), int((delta == 1).sum()))


if __name__ == "__main__":
    main()
