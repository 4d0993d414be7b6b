#!/usr/bin/env python
"""bench.py -- stabilized frames/sec of the LiveVisionKit stabilization hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 the driver launches it through
torch.distributed.run, one rank per GPU).  A "step" is one pass of the hot path over one frame in steady state
(one lvk_hip_stab_push: track the new frame, smooth the path, remap the delayed frame).  Independent streams shard
one per GPU with no data-path collective; the only torch.distributed calls are the barrier
and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.

Workload (config.workload): one 3840x2160 packed-YUV stream per GPU, frames already resident in HBM, OBS
"Homography" preset (tracking 480x270, 2x1 regions, 2x2 mesh), predictive_samples = 10, auto-crop 5 %.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)     # 2000 x ~0.14 ms: a 0.3 s timed region (a 40 ms one is at the mercy of clock ramps)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--rows", type=int, default=2160)
    ap.add_argument("--cols", type=int, default=3840)
    ap.add_argument("--preset", default="homography", choices=["homography", "field"])
    ap.add_argument("--pool", type=int, default=24, help="distinct source frames kept in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--format", default="i420", choices=["i420", "nv12", "packed"],
                    help="frame format resident in HBM: 4:2:0 planes in and out (BASELINE metric) or the packed 8UC3 boundary format")
    ap.add_argument("--lens", default="off", choices=["off", "fused", "two-pass"],
                    help="BASELINE config 5: lens-correction pre-warp (profile of SURVEY.md section 8d) fused into the stabilizing remap, "
                         "or as the reference chain's separate LC pass (two-pass; --format packed only)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the extra (untimed-for-`value`) pass with host-resident frames")
    ap.add_argument("--no-overlap", action="store_true", help="keep the output remap on the tracking stream")
    ap.add_argument("--cpu-frames", type=int, default=0, help="CPU baseline: one pass over the frame pool instead of a 12 s budget")
    return ap.parse_args()


def make_frame_pool(rows, cols, count, seed, device):
    """Synthetic shaky stream generated on the GPU: one textured canvas (gratings + rectangles + noise), `count`
    frames cropped at jittered integer offsets (smooth pan + AR(1) jitter)."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    m = int(0.04 * cols) + 8
    H, W = rows + 2 * m, cols + 2 * m
    yy = torch.arange(H, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    img = torch.full((H, W), 128.0, device=device)
    for _ in range(12):
        th, f, ph, a = rng.uniform(0, np.pi), rng.uniform(0.004, 0.06), rng.uniform(0, 6.28), rng.uniform(4, 14)
        img += a * torch.sin((np.cos(th) * xx + np.sin(th) * yy) * (f * 6.2832) + ph)
    nrect = 2500
    ys = rng.integers(0, H - 8, nrect); xs = rng.integers(0, W - 8, nrect)
    hs = rng.integers(8, max(9, H // 10), nrect); ws = rng.integers(8, max(9, W // 10), nrect)
    vs = rng.uniform(10, 245, nrect)
    for i in range(nrect):
        img[ys[i]:ys[i] + hs[i], xs[i]:xs[i] + ws[i]] = float(vs[i])
    img += torch.randn((H, W), device=device, generator=g) * 1.5
    canvas = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
    canvas[..., 0] = img.clamp(0, 255).to(torch.uint8)
    canvas[..., 1] = (128 + 60 * torch.sin(xx / W * 5.0 + 0.3) + 20 * torch.cos(yy / H * 7.0)).clamp(0, 255).to(torch.uint8)
    canvas[..., 2] = (128 + 50 * torch.cos(xx / W * 3.0 - yy / H * 4.0)).clamp(0, 255).to(torch.uint8)
    del img
    ar = np.zeros(2); offs = []
    for i in range(count):
        ar = 0.6 * ar + 0.8 * rng.normal(0, 1, 2) * 0.004 * cols
        # closed pan loop so that the pool can be cycled without a discontinuity larger than the jitter
        pan = 0.25 * m * np.array([np.sin(2 * np.pi * i / count), np.cos(2 * np.pi * i / count)])
        o = np.clip(np.rint(ar + pan), -m + 1, m - 1).astype(int)
        offs.append(o)
    frames = [canvas[m + o[1]:m + o[1] + rows, m + o[0]:m + o[0] + cols].contiguous() for o in offs]
    return frames


def cpu_baseline(rows, cols, preset_name, frames_host, nthreads, budget_s=12.0, fmt="packed", lens_params=None):
    """The CPU oracle (a port: CPU restatement of the reference, see oracle/lvk_oracle.h) timed on the host cores on a
    bounded sample of the same workload."""
    from tests import oracle_lib
    oracle = oracle_lib.load()
    s = oracle_lib.preset(preset_name)
    st = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default"))      # same OBS flow as the GPU leg
    st.configure(s)
    if lens_params is not None:
        st.set_lens(lens_params)
    delay = s.predictive_samples
    n = len(frames_host)
    yuv420 = fmt != "packed"
    if yuv420:
        planes_host = [oracle.egress_yuv420(f, nv12=(fmt == "nv12")) for f in frames_host]

    def one(i):
        if not yuv420:
            return st.push(frames_host[i % n], ts=i, nthreads=nthreads)
        packed = oracle.ingest_yuv420(*planes_host[i % n])              # ingest -> filter -> egress, as the GPU path
        out, ts = st.push(packed, ts=i, nthreads=nthreads)
        if out is not None:
            oracle.egress_yuv420(out, nv12=(fmt == "nv12"))
        return out, ts

    # untimed: build the delay with cycled frames
    for i in range(delay + 1):
        one(i)
    t0 = time.perf_counter()
    done = 0
    i = 0
    while True:
        out, _ = one(delay + 1 + i)
        done += 1 if out is not None else 0
        i += 1
        dt = time.perf_counter() - t0
        if (budget_s and dt >= budget_s) or (not budget_s and i >= n):
            break
    st.close()
    return done / dt, dt, done


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    import livevisionkit_amd as lvk
    numa_cpus = lvk.shard.bind_to_gpu_numa(local_rank) if os.environ.get("LVK_BENCH_NUMA", "1") != "0" else []
    # the filter works on its own (non-blocking) stream: the process default stream would implicitly serialise with every
    # blocking stream of the process
    work_stream = torch.cuda.Stream(device, priority=int(os.environ.get("LVK_BENCH_STREAM_PRIO", "0")))
    ctx = lvk.Context(local_rank, stream=work_stream)
    settings = lvk.StabilizationFilterSettings.obs_preset(args.preset)
    # the OBS plugin's flow (VSFilter.cpp:255-293): a default-constructed filter that is then configured with the preset --
    # constructing the "field" preset directly would keep FrameTracker's constructor-time 256x256 mesh constraints (reference quirk)
    filt = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx)
    filt.configure(settings)
    if not args.no_overlap:
        filt.set_overlap(True)          # remap of frame n-N on a second stream, concurrent with the tracking of frame n+1
    delay = filt.frame_delay()

    rows, cols = args.rows, args.cols
    lens_params = (0.8 * cols, 0.8 * cols, cols / 2, rows / 2, -0.12, 0.03, 0.0, 0.0, 0.0)
    lens_map = lens_bufs = None
    if args.lens == "fused":
        filt.set_lens(lens_params)
    elif args.lens == "two-pass":
        if args.format != "packed":
            raise SystemExit("--lens two-pass needs --format packed")
        lens_map, _ = ctx.lens_map(lens_params, rows, cols)
    pool = max(args.pool, delay + 3)
    frames = make_frame_pool(rows, cols, pool, seed=0x4C564B31 + rank, device=device)
    yuv420 = args.format != "packed"
    if yuv420:
        # convert the synthetic stream to 4:2:0 planes once (outside the timed region) and drop the packed copies
        planes = [ctx.egress_yuv420(f, nv12=(args.format == "nv12")) for f in frames]
        ctx.sync()
        host_packed = None
        outs = [tuple(torch.empty_like(p) for p in planes[0]) for _ in range(4)]
        planes_args = [filt.prepare_yuv420(p) for p in planes]          # addresses / pitches marshalled once, outside the timed region
        outs_args = [filt.prepare_yuv420(o) for o in outs]
    else:
        outs = [torch.empty_like(frames[0]) for _ in range(4)]
    if lens_map is not None:
        lens_bufs = [torch.empty_like(frames[0]) for _ in range(delay + 4)]       # corrected frames stay borrowed for `delay` pushes
    torch.cuda.synchronize()

    step_no = [0]

    def step():
        i = step_no[0]; step_no[0] += 1
        if yuv420:
            return filt.apply_yuv420_prepared(planes_args[i % pool], i, outs_args[i & 3])
        if lens_map is not None:
            corrected = ctx.remap_map(frames[i % pool], lens_map, bg=(0, 0, 0), out=lens_bufs[i % len(lens_bufs)])     # LCFilter::filter
            return filt.apply(corrected, timestamp=i, out=outs[i & 3])
        return filt.apply(frames[i % pool], timestamp=i, out=outs[i & 3])

    # fill the delay (untimed, before the warmup): every timed step then emits one stabilized frame
    for _ in range(delay + 2):
        step()
    for _ in range(args.warmup):
        step()
    # live HIP-event timing of the dominant kernel inside the timed region: that stage only, and one launch in eight (an event pair per
    # frame is two host API calls in the per-frame turnaround -- the measurement would slow what it measures by ~4 %)
    filt.set_profiling(True, stages=("remap",), every=8)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    emitted = 0
    stamps = [t0]
    for _ in range(args.steps):
        out, _ = step()
        emitted += 1 if out is not None else 0
        stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    free_running = np.diff(np.array(stamps)) * 1e3          # host time per push in the free-running timed region
    barrier()
    prof = filt.profile()
    filt.set_profiling(False)
    stats = filt.stats()

    # per-step latency pass (each step synchronised) for p50 / p99 ms per frame
    lat = []
    for _ in range(min(args.steps, 500)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t) * 1e3)

    # every stage's event timing in a separate free-running pass (outside the timed region: 16 event records per frame)
    filt.set_profiling(True)
    for _ in range(min(args.steps, 200)):
        step()
    torch.cuda.synchronize()
    prof_all = filt.profile()
    filt.set_profiling(False)

    # the same remap kernel alone on the GPU at full occupancy (the timed region runs its occupancy-capped `_co` variant next to the tracker)
    standalone_us = None
    if rank == 0 and args.lens != "two-pass":
        torch.cuda.synchronize()
        meshes = filt.meshes()[1]
        src = frames[0]; dst = torch.empty_like(src)
        bgc = tuple(int(v) for v in settings.background)
        with torch.cuda.stream(work_stream):
            for _ in range(3):
                ctx.warpmesh_apply(src, meshes, bg=bgc, yuv=True, out=dst)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(work_stream)
            for i in range(20):
                ctx.warpmesh_apply(frames[i % pool], meshes, bg=bgc, yuv=True, out=dst)
            e1.record(work_stream)
        torch.cuda.synchronize()
        standalone_us = e0.elapsed_time(e1) / 20 * 1e3

    # PCIe-inclusive rate (reported beside `value`, never as it): the same stream with the I420 planes in pinned HOST memory --
    # H2D of frame i + 1 on an upload stream while frame i is processed, D2H of every output chained behind its remap on the
    # filter's output stream (lvk_hip_stab_output_stream), no host synchronisation inside the loop
    pcie = None
    if rank == 0 and yuv420 and not args.no_pcie and not args.no_overlap:
        try:
            nsteps = min(args.steps, 1000)
            host_in = [tuple(p.cpu().pin_memory() for p in pl) for pl in planes]
            K = 4
            dev_in = [tuple(torch.empty_like(p) for p in planes[0]) for _ in range(K)]
            dev_in_args = [filt.prepare_yuv420(d) for d in dev_in]
            host_out = [tuple(torch.empty_like(p, device="cpu").pin_memory() for p in planes[0]) for _ in range(4)]
            up = torch.cuda.Stream(device); down = torch.cuda.Stream(device)
            out_stream = filt.output_stream()
            ev_up = [torch.cuda.Event() for _ in range(K)]
            ev_out = [torch.cuda.Event() for _ in range(4)]; ev_down = [None] * 4

            def upload(i):
                with torch.cuda.stream(up):
                    for d, h in zip(dev_in[i % K], host_in[i % pool]):
                        d.copy_(h, non_blocking=True)
                    ev_up[i % K].record(up)

            def run(n, base):
                upload(base)
                for i in range(base, base + n):
                    upload(i + 1)
                    work_stream.wait_event(ev_up[i % K]); out_stream.wait_event(ev_up[i % K])
                    if ev_down[i & 3] is not None:
                        out_stream.wait_event(ev_down[i & 3])          # the output planes of 4 pushes ago have left the device
                    res, _ = filt.apply_yuv420_prepared(dev_in_args[i % K], i, outs_args[i & 3])
                    if res is not None:
                        ev_out[i & 3].record(out_stream)
                        down.wait_event(ev_out[i & 3])
                        with torch.cuda.stream(down):                  # downloads run beside the next frame's remap
                            for h, d in zip(host_out[i & 3], outs[i & 3]):
                                h.copy_(d, non_blocking=True)
                            ev_down[i & 3] = torch.cuda.Event(); ev_down[i & 3].record(down)
            torch.cuda.synchronize()
            run(100, step_no[0]); step_no[0] += 100
            torch.cuda.synchronize()
            tp = time.perf_counter()
            run(nsteps, step_no[0]); step_no[0] += nsteps
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp
            mb = sum(p.numel() for p in planes[0]) / 1e6
            # per-frame latency with the transfers inside (BASELINE's p99 ms/frame for host-resident frames): upload the planes, push,
            # download the emitted planes, synchronise -- one frame at a time, nothing prefetched
            lat_pcie = []
            for i in range(step_no[0], step_no[0] + 200):
                tl = time.perf_counter()
                upload(i)
                work_stream.wait_event(ev_up[i % K]); out_stream.wait_event(ev_up[i % K])
                res, _ = filt.apply_yuv420_prepared(dev_in_args[i % K], i, outs_args[i & 3])
                if res is not None:
                    ev_out[i & 3].record(out_stream)
                    down.wait_event(ev_out[i & 3])
                    with torch.cuda.stream(down):
                        for h, d in zip(host_out[i & 3], outs[i & 3]):
                            h.copy_(d, non_blocking=True)
                torch.cuda.synchronize()
                lat_pcie.append((time.perf_counter() - tl) * 1e3)
            step_no[0] += 200
            pcie = {"value": nsteps / dtp, "unit": "frames/s", "host_to_device_MB_per_frame": mb, "device_to_host_MB_per_frame": mb,
                    "GBps_each_way": nsteps / dtp * mb / 1e3,
                    "latency_ms": {"p50": float(np.percentile(lat_pcie, 50)), "p99": float(np.percentile(lat_pcie, 99)),
                                   "note": "upload + push + download + synchronise, one frame at a time"},
                    "note": "same stream, I420 planes in pinned host memory; uploads prefetched one frame ahead on their own stream, downloads on a third "
                            "stream behind an event on the filter's output stream; not the headline value (inputs of `value` are resident in HBM)"}
        except Exception as e:          # the extra pass must never break the contract line
            pcie = {"error": repr(e)}

    elapsed_max, total_frames = lvk.shard.reduce_timing(elapsed, emitted, device=device)

    result = None
    if rank == 0:
        remap_ms, remap_n = prof["remap"]
        # S_in + S_out of the frames the dominant kernel reads / writes as stored on the device: the packed 8UC3
        # frame in; out = packed 8UC3 (6 W H in total) or, on the 4:2:0 path, the planes of the fused remap + egress kernel (4.5 W H)
        fused_420 = yuv420 and args.lens != "two-pass"
        alg_bytes = (9 * rows * cols) // 2 if fused_420 else 6 * rows * cols
        achieved = (alg_bytes / (remap_ms / remap_n * 1e-3)) / 1e9 if remap_n else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "remap_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                if t.get("rows") == rows and t.get("cols") == cols and ("_420" in t.get("kernel", "")) == fused_420:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "stabilized frames/sec (one 4K YUV420 stream per GPU, steady state)" if yuv420 else
                      "stabilized frames/sec (one 4K packed-YUV444 stream per GPU, steady state)",
            "value": total_frames / elapsed_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"{cols}x{rows} {args.format.upper()} (YUV 4:2:0) stream per GPU, planes resident in HBM, ingest -> lvk::StabilizationFilter -> egress, "
                                    if yuv420 else f"{cols}x{rows} packed YUV444 8UC3 stream per GPU (lvk::StabilizationFilter boundary format), ")
                                   + f"OBS '{args.preset}' preset, tracking 480x270, predictive_samples={delay}, crop 5%"
                                   + ("" if args.lens == "off" else f", lens correction {args.lens} (fx=fy=0.8W, k1=-0.12, k2=0.03)"),
                       "parallelism": f"{world} independent stream(s), one per GPU, no collective",
                       "frames_in_hbm": pool, "host_cpus_bound": len(numa_cpus)},
            "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99))},
            "free_running_ms": {"p10": float(np.percentile(free_running, 10)), "p50": float(np.percentile(free_running, 50)),
                                "p90": float(np.percentile(free_running, 90)), "p99": float(np.percentile(free_running, 99))},
            "stage_us": {k: (v[0] / v[1] * 1e3 if v[1] else 0.0) for k, v in prof_all.items()},
            "pcie_inclusive": pcie,
            "tracking": {"stability": stats.tracking_stability, "trust": stats.trust, "features": stats.n_tracked},
            "roofline": {"kernel": ("k_remap_homography" if args.preset == "homography" else "k_remap_mesh") + ("_lens" if args.lens == "fused" else "")
                                   + (("_420<nv12>" if args.format == "nv12" else "_420<i420>") if fused_420 else ("<yuv>" if args.no_overlap else "_co<yuv>")),
                         "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": remap_ms / remap_n * 1e3 if remap_n else None, "launches": remap_n,
                         # the full-occupancy kernel alone on the GPU (outside the timed region), same frames and warp
                         "standalone_us": standalone_us,
                         "standalone_frac": (6 * rows * cols / (standalone_us * 1e-6)) / 1e9 / 8000.0 if standalone_us else None,
                         # the kernel is VALU-issue bound: 531 (packed output) / 534 (planar 4:2:0 output) VALU wave-instructions per output
                         # pixel (rocprofv3 SQ_INSTS_VALU, profiles/r01_sq_counters_per_kernel.txt) against 64.6 T lane-instr/s measured
                         # with scripts/valu_peak.hip
                         "valu_frac": ((534.0 if fused_420 else 531.0) * rows * cols / (remap_ms / remap_n * 1e-3)) / 64.6e12 if remap_n else None,
                         "standalone_valu_frac": (531.0 * rows * cols / (standalone_us * 1e-6)) / 64.6e12 if standalone_us else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            ncpu = os.cpu_count() or 1
            nthreads = min(ncpu, 64)
            host_frames = [f.cpu().numpy() for f in frames]
            fps, dt, done = cpu_baseline(rows, cols, args.preset, host_frames, nthreads,
                                         budget_s=0.0 if args.cpu_frames else 12.0, fmt=args.format,
                                         lens_params=lens_params if args.lens == "fused" else None)
            result["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": nthreads, "kind": "port",
                                      "sample": f"{done} steady-state frames of the same workload ({dt:.1f} s of CPU work; "
                                                f"oracle = CPU restatement of the reference, remap row-parallel over {nthreads} threads, "
                                                f"tracker and 4:2:0 conversion single-threaded)"}
        print(json.dumps(result), flush=True)
    filt.close()
    if world > 1:
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
