"""Micro-benchmark of lvk_hip_ingest_obs / lvk_hip_egress_obs per OBS video format (HIP events on the context's stream).  usage: python scripts/bench_ingest_obs.py [rows cols]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import livevisionkit_amd as lvk  # noqa: E402


def shapes(fmt, rows, cols):
    return {"I420": [(rows, cols), (rows // 2, cols // 2), (rows // 2, cols // 2)], "NV12": [(rows, cols), (rows // 2, cols // 2, 2)],
            "I422": [(rows, cols), (rows, cols // 2), (rows, cols // 2)], "I444": [(rows, cols)] * 3, "YUY2": [(rows, cols, 2)], "UYVY": [(rows, cols, 2)],
            "YVYU": [(rows, cols, 2)], "AYUV": [(rows, cols, 4)], "BGR3": [(rows, cols, 3)], "RGBA": [(rows, cols, 4)]}[fmt]


def main():
    rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    stream = torch.cuda.Stream()
    ctx = lvk.Context(0, stream=stream)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    with torch.cuda.stream(stream):
        for fmt in ("I420", "NV12", "I422", "I444", "YUY2", "UYVY", "YVYU", "AYUV", "BGR3", "RGBA"):
            sets = [[torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g) for sh in shapes(fmt, rows, cols)] for _ in range(6)]
            frame = torch.empty((rows, cols, 3), dtype=torch.uint8, device="cuda")
            res = {}
            for name, fn in (("ingest", lambda p: ctx.ingest_obs(fmt, p, out=frame)), ("egress", lambda p: ctx.egress_obs(fmt, frame, p))):
                for p in sets[:2]:
                    fn(p)
                ctx.sync()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = 40
                e0.record(stream)
                for i in range(iters):
                    fn(sets[i % len(sets)])
                e1.record(stream)
                ctx.sync(); torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) / iters * 1e3
            byts = sum(int(torch.tensor(sh).prod()) for sh in shapes(fmt, rows, cols)) + rows * cols * 3
            print(f"{fmt:5s} {cols}x{rows}: ingest {res['ingest']:7.1f} us ({byts / res['ingest'] / 1e3:6.2f} GB/s... {byts / res['ingest'] * 1e6 / 8e12 * 100:4.1f} % of 8 TB/s)   egress {res['egress']:7.1f} us ({byts / res['egress'] * 1e6 / 8e12 * 100:4.1f} %)")


if __name__ == "__main__":
    main()
