#!/usr/bin/env python
"""bench.py -- stabilized frames/sec of the LiveVisionKit stabilization hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 the driver launches it through torch.distributed.run, one rank per
GPU).  A "step" is one pass of the hot path over one BATCH of synthetic input: 16 consecutive frames of the stream in steady state (16 x
lvk_hip_stab_push_yuv420: ingest the 4:2:0 planes, track the new frame, smooth the path, remap + egress the delayed frame; --frames-per-step).
Rounds 1-5 stepped one frame at a time; the driver's `--steps 20` region was then 2.5 ms between two device-wide synchronisations and read 6 800-8 800
frames/s from run to run on one box while the same line's 600-frame continuation read 8 570-9 200 -- 20 batches are 35 ms and agree with it to ~1 %.  Independent streams shard one per GPU with no data-path collective
(SURVEY.md section 8e): the process group is gloo (CPU) and carries only the barrier and the max-over-ranks of the elapsed time -- no RCCL.
Rank 0 prints ONE JSON line.

Workload (config.workload): one 3840x2160 I420 stream per GPU, planes resident in HBM when the timed region starts; SURVEY.md section
8d's synthetic clip (tests/clipgen.py: 600 distinct camera poses -- smooth pan + AR(1) jitter in translation / rotation / zoom --, a scene
cut at frame 300, ground-truth homographies and the jitter-free render); OBS "Homography" preset (tracking 480x270, 2x1 regions, 2x2
mesh), predictive_samples = 10, auto-crop 5 %.  Beside `value`: latency (always >= 500 synchronised pushes), the live roofline of the remap
(always >= 64 HIP-event samples inside / right after the timed region), the PCIe-inclusive rate, the picture quality against the ideal
render for the GPU and for the CPU oracle, the oracle timed on the host cores, the reference's own remap kernel (compiled for this chip)
timed on the same frame, short legs of the other BASELINE configurations, and K concurrent streams on this one GPU.

`--gpus N` with N > 1 launched plainly (no torch.distributed.run around it) spawns its N ranks itself; whichever way it was launched, the
line's `n_gpus` is the number of ranks that timed, and the run fails (exit code 3) when that is not --gpus."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md
VALU_PEAK_SPEC = 78.6e12              # fp32 lane-instructions / s at the 2.4 GHz peak clock: 157.3 TFLOP/s vector peak / 2 (an fma counts as two flops)
# the committed counter summaries the roofline block reads (tests/test_bench_profiles.py: every key read from them exists)
PROFILE_FILES = {"traffic": ("remap_pmc_traffic.json", "remap_pmc_traffic_field.json"), "stalls": "remap_stalls.json"}
PROFILE_KEYS = {"traffic": ("rows", "cols", "kernel", "hbm_bytes_per_launch", "valu_per_px", "valu_per_px_packed"),
                "stalls": ("clock_mhz_under_kernel", "valu_issue_slot_frac", "active_inst_any_frac", "wait_inst_any_frac", "wait_any_frac", "waves_per_simd")}
SPEC_CLOCK_MHZ = 2400.0               # 256 CUs x 4 SIMDs x 32 lanes per cycle x 2.4 GHz = 78.6 T lane-instructions / s


def _gpu_sysfs_dir(pci_bus_id=None):
    """/sys/class/drm/cardN/device of the GPU (matched by PCI bus id when given, else the first amdgpu card with a clock file)."""
    import glob
    cands = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    for d in cands:
        try:
            if pci_bus_id and pci_bus_id.lower() in os.path.realpath(d).lower():
                return d
        except OSError:
            pass
    for d in cands:
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")) or os.path.exists(os.path.join(d, "gpu_busy_percent")):
            return d
    return None


def _read_gpu_sensors(d):
    """(shader clock MHz, socket power W) from the amdgpu sysfs / hwmon files; None where the file is missing."""
    import glob
    mhz = watts = None
    for f in glob.glob(os.path.join(d, "hwmon", "hwmon*", "freq1_input")):
        try:
            mhz = int(open(f).read()) / 1e6; break
        except (OSError, ValueError):
            pass
    if mhz is None:
        try:
            for line in open(os.path.join(d, "pp_dpm_sclk")):
                if "*" in line:
                    mhz = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
    for name in ("power1_average", "power1_input"):
        for f in glob.glob(os.path.join(d, "hwmon", "hwmon*", name)):
            try:
                watts = int(open(f).read()) / 1e6; break
            except (OSError, ValueError):
                pass
        if watts is not None:
            break
    return mhz, watts


def sampler_main(argv):
    """`bench.py --gpu-sampler <sysfs dir> <seconds> <period ms>`: a process of its own (no GIL shared with the timed loop) that samples the
    GPU's shader clock and socket power and prints one JSON list of [unix time, MHz, W] when told to stop (a line on stdin) or timed out."""
    d, seconds, period = argv[0], float(argv[1]), float(argv[2]) / 1e3
    out, end = [], time.time() + seconds
    import select
    while time.time() < end:
        mhz, watts = _read_gpu_sensors(d)
        out.append([time.time(), mhz, watts])
        if select.select([sys.stdin], [], [], period)[0]:
            break
    print(json.dumps(out), flush=True)


class GpuSampler:
    def __init__(self, pci_bus_id=None, seconds=60.0, period_ms=2.0):
        import subprocess
        self.proc = None
        d = _gpu_sysfs_dir(pci_bus_id)
        if d is None:
            return
        self.dir = d
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gpu-sampler", d, str(seconds), str(period_ms)],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def stop(self):
        if self.proc is None:
            return []
        try:
            out, _ = self.proc.communicate("stop\n", timeout=10)
            return json.loads(out.strip().splitlines()[-1])
        except Exception:
            return []

    @staticmethod
    def window(samples, t0, t1):
        """mean / min / max clock and mean power of the samples taken inside [t0, t1] (unix time)."""
        sel = [s for s in samples if t0 <= s[0] <= t1]
        mhz = [s[1] for s in sel if s[1] is not None]; w = [s[2] for s in sel if s[2] is not None]
        if not sel:
            return None
        return {"samples": len(sel), "clock_mhz": {"mean": float(np.mean(mhz)), "min": float(np.min(mhz)), "max": float(np.max(mhz))} if mhz else None,
                "power_w": float(np.mean(w)) if w else None}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)      # 250 steps x 16 frames x ~0.11 ms: a 0.45 s timed region
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--frames-per-step", type=int, default=16,
                    help="one step = one batch of this many consecutive frames of the stream (that many pushes).  Round 6: a step of ONE frame made the driver's "
                         "`--steps 20` region 2.5 ms between two device-wide synchronisations, 8-12 %% of which are the empty pipeline behind the first and the "
                         "lone remap in front of the second (6 800-8 800 frames/s from run to run on one box); 20 batches of 16 frames are 35 ms")
    ap.add_argument("--rows", type=int, default=2160)
    ap.add_argument("--cols", type=int, default=3840)
    ap.add_argument("--preset", default="homography", choices=["homography", "field"])
    ap.add_argument("--pool", type=int, default=600, help="distinct source frames (camera poses) kept in HBM; scene cut at pool / 2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--format", default="i420", choices=["i420", "nv12", "packed"],
                    help="frame format resident in HBM: 4:2:0 planes in and out (BASELINE metric) or the packed 8UC3 boundary format")
    ap.add_argument("--lens", default="off", choices=["off", "fused", "two-pass"],
                    help="BASELINE config 5: lens-correction pre-warp (profile of SURVEY.md section 8d) fused into the stabilizing remap, "
                         "or as the reference chain's separate LC pass (two-pass; --format packed only)")
    ap.add_argument("--no-lookahead", action="store_true", help="skip the extra pass in which every frame is announced one push ahead")
    ap.add_argument("--no-pcie", action="store_true", help="skip the extra (untimed-for-`value`) pass with host-resident frames")
    ap.add_argument("--no-overlap", action="store_true", help="keep the output remap on the tracking stream")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work of the oracle baseline")
    ap.add_argument("--quality-frames", type=int, default=90, help="frames of the picture-quality pass (GPU and oracle vs the ideal render); 0 = skip")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="K independent streams (filters, HIP streams, clips, host threads) on every GPU: `value` is the aggregate over all of them. "
                         "One 4K60 stream occupies < 1 %% of the GPU -- streams per device is the scaling that matters to a deployment")
    ap.add_argument("--input", default=None, help="raw 4:2:0 file (I420, or NV12 with --format nv12) of --cols x --rows frames instead of the synthetic clip "
                                                  "(the reference's harness reads a clip: Modules/VideoEditor/VideoProcessor.cpp:148-230)")
    ap.add_argument("--write-input", default=None, help="write the first --pool frames of the synthetic clip to this file (raw I420 / NV12) and exit")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other BASELINE configurations (1080p, lens-fused, field preset)")
    ap.add_argument("--no-reference-kernel", action="store_true", help="skip timing the reference's compiled remap kernel (oracle/_ref) on the same frame")
    ap.add_argument("--no-multi-stream", action="store_true", help="skip the extra leg with 4 concurrent streams on this GPU")
    ap.add_argument("--no-sensors", action="store_true", help="do not sample the GPU's clock / power sensors (sysfs) during the timed regions")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` launched plainly: run the N ranks through torch.distributed.run ourselves (the way the driver launches N > 1),
    check the one line they print against --gpus, and pass it on.  A multi-GPU run can then not silently degrade to one rank."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        sys.stdout.write(p.stdout)
        raise SystemExit(p.returncode if p.returncode else 3)
    r = json.loads(lines[0])
    if r.get("n_gpus") != args.gpus or len(r.get("ranks", [])) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {r.get('n_gpus')} rank(s) timed\n")
        raise SystemExit(3)
    print(lines[0], flush=True)
    return r


def percentiles(x, ps=(50, 99)):
    return dict({"p%d" % p: float(np.percentile(x, p)) for p in ps}, max=float(np.max(x)))


def cpu_baseline(oracle, rig, preset_name, nthreads, budget_s, fmt, lens_params, delay, check_against=None):
    """The CPU oracle (a port: CPU restatement of the reference, oracle/lvk_oracle.h) timed on the host cores on a bounded sample of the
    same workload: ingest -> filter -> egress of consecutive frames of the same clip, every stage row / point-parallel.  BASELINE.md section 3:
    frames/s plus p50 / p99 ms per frame plus per-stage ms (downscale, detect, LK, estimate, smooth, remap -- the oracle's own timers -- and
    the 4:2:0 conversions either side), at T = `nthreads` (one per physical core) for `budget_s` seconds and then at T = 1 for budget_s / 2.
    check_against: the parity oracle -- the first emitted frames of `oracle` (the -O3 -march=native timing build) must equal its frames."""
    from tests import oracle_lib
    s = oracle_lib.preset(preset_name)
    st = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default"))      # same OBS flow as the GPU leg
    st.configure(s)
    ref = None
    if check_against is not None:
        ref = oracle_lib.OracleStabilizer(check_against, oracle_lib.preset("default")); ref.configure(s)
    if lens_params is not None:
        st.set_lens(lens_params)
        if ref is not None:
            ref.set_lens(lens_params)
    yuv420 = fmt != "packed"
    nv12 = fmt == "nv12"
    oracle.set_num_threads(nthreads)
    conv_ms = {"ingest": 0.0, "egress": 0.0}

    def source(i):
        if rig.clip is None:                                 # --input: the file's own planes
            return tuple(np.ascontiguousarray(q) for q in rig.file_planes(i))
        f = rig.clip.render444(i).cpu().numpy()
        return oracle.egress_yuv420(f, nv12=nv12) if yuv420 else f

    def one(i, src, stab, lib, threads, timed=True):
        if not yuv420:
            return stab.push(src, ts=i, nthreads=threads)
        t0 = time.perf_counter()
        packed = lib.ingest_yuv420(*src)                     # ingest -> filter -> egress, as the GPU path
        t1 = time.perf_counter()
        out, ts = stab.push(packed, ts=i, nthreads=threads)
        t2 = time.perf_counter()
        if out is not None:
            out = lib.egress_yuv420(out, nv12=nv12)
        if timed:
            conv_ms["ingest"] += (t1 - t0) * 1e3; conv_ms["egress"] += (time.perf_counter() - t2) * 1e3
        return out, ts

    same = None
    for i in range(delay + 1):                               # untimed: build the delay
        src = source(i)
        one(i, src, st, oracle, nthreads, timed=False)
        if ref is not None:
            one(i, src, ref, check_against, nthreads, timed=False)
    i = delay + 1
    if ref is not None:                                      # the timing build must be the same function: two emitted frames, bit for bit
        same = True
        for _ in range(2):
            src = source(i)
            a, _ = one(i, src, st, oracle, nthreads, timed=False); b_, _ = one(i, src, ref, check_against, nthreads, timed=False)
            a = a if isinstance(a, tuple) else (a,); b_ = b_ if isinstance(b_, tuple) else (b_,)
            same = same and all(np.array_equal(x, y) for x, y in zip(a, b_))
            i += 1
        ref.close()

    def leg(threads, budget):
        nonlocal i
        oracle.set_num_threads(threads)
        st.stage_ms(reset=True); conv_ms["ingest"] = conv_ms["egress"] = 0.0
        done, spent, per = 0, 0.0, []
        while spent < budget or done < 3:
            src = source(i)                                  # rendering the synthetic frame is not part of the workload
            t0 = time.perf_counter()
            out, _ = one(i, src, st, oracle, threads)
            dt = time.perf_counter() - t0
            spent += dt; per.append(dt * 1e3)
            done += 1 if out is not None else 0
            i += 1
        stages = {k: v / len(per) for k, v in st.stage_ms().items()}
        stages.update({k: v / len(per) for k, v in conv_ms.items()})
        return {"value": done / spent, "unit": "frames/s", "cores": threads, "frames": done, "cpu_seconds": spent,
                "p50_ms": float(np.percentile(per, 50)), "p99_ms": float(np.percentile(per, 99)), "stage_ms": stages}
    full = leg(nthreads, budget_s)
    single = leg(1, budget_s / 2.0) if nthreads > 1 else None
    st.close()
    oracle.set_num_threads(1)
    return full, single, same


def quality_pass(lvk, ctx, oracle, clip, preset_name, nframes, nthreads):
    """End-to-end picture quality: the stabilized frames of the first `nframes` of the clip against the
    jitter-free ideal render, for the HIP path and for the CPU oracle (crop_to_stable_region off so that output and ideal share one
    geometry; relaxed QA so that the trust factor is up within the pass; luma PSNR over the central 84 % of the frame)."""
    import torch
    from tests import clipgen, oracle_lib
    over = dict(crop_to_stable_region=0, min_scene_quality=0.4, min_tracking_quality=0.2)
    so = oracle_lib.preset(preset_name, **over)
    sg = lvk.StabilizationFilterSettings.obs_preset(preset_name, strict=False, apply_crop=False)
    gst = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=ctx); gst.configure(sg)
    ost = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); ost.configure(so)
    delay = sg.predictive_samples
    first = min(nframes - 1, 3 * delay)                      # the smoother's window is full and the trust factor has ramped up
    res = {"frames": 0, "psnr_unstabilized": [], "psnr_gpu": [], "psnr_oracle": [], "gpu_equals_oracle": True}
    for i in range(nframes):
        f = clip.render444(i)
        torch.cuda.synchronize()                             # rendered on torch's current stream, consumed on the filter's
        got, gts = gst.apply(f, timestamp=i)
        ctx.sync()
        want, wts = ost.push(f.cpu().numpy(), ts=i, nthreads=nthreads)
        assert (got is None) == (want is None)
        if got is None or wts < first:
            continue
        ideal = clip.render444(wts, smooth=True)
        res["psnr_unstabilized"].append(clipgen.psnr_region(clip.render444(wts), ideal))
        res["psnr_gpu"].append(clipgen.psnr_region(got, ideal))
        res["psnr_oracle"].append(clipgen.psnr_region(torch.from_numpy(want).to(ideal.device), ideal))
        res["gpu_equals_oracle"] = res["gpu_equals_oracle"] and bool(np.array_equal(got.cpu().numpy(), want))
        res["frames"] += 1
    trust = gst.stats().trust
    gst.close(); ost.close()
    out = {k: (float(np.mean(v)) if isinstance(v, list) and v else v) for k, v in res.items()}
    out["trust_at_end"] = float(trust)
    out["note"] = (f"mean luma PSNR (dB) against the jitter-free render over the central 84 % of the frame, frames {first}..{nframes - 1 - delay} "
                   "of the clip, crop_to_stable_region off, relaxed QA")
    return out


class Rig:
    """One stream: a context on a HIP stream of its own, a filter, its frames resident in HBM, and step() = one push in steady state."""

    def __init__(self, lvk, local_rank, device, seed, rows, cols, preset, fmt, lens, overlap, pool, input_path=None, cut=True, pingpong=False):
        import torch
        from tests import clipgen
        self.lvk, self.rows, self.cols, self.fmt, self.lens, self.preset, self.seed = lvk, rows, cols, fmt, lens, preset, seed
        # the filter works on its own (non-blocking) stream: the process default stream would implicitly serialise with every
        # blocking stream of the process
        self.stream = torch.cuda.Stream(device)
        self.ctx = lvk.Context(local_rank, stream=self.stream)
        self.settings = lvk.StabilizationFilterSettings.obs_preset(preset)
        # the OBS plugin's flow (VSFilter.cpp:255-293): a default-constructed filter that is then configured with the preset --
        # constructing the "field" preset directly would keep FrameTracker's constructor-time 256x256 mesh constraints (reference quirk)
        self.filt = lvk.StabilizationFilter(lvk.StabilizationFilterSettings(), context=self.ctx)
        self.filt.configure(self.settings)
        if overlap:
            self.filt.set_overlap(True)          # remap of frame n-N on a second stream, concurrent with the tracking of frame n+1
        self.delay = self.filt.frame_delay()
        self.lens_params = (0.8 * cols, 0.8 * cols, cols / 2, rows / 2, -0.12, 0.03, 0.0, 0.0, 0.0)
        self.lens_map = self.lens_bufs = None
        if lens == "fused":
            self.filt.set_lens(self.lens_params)
        elif lens == "two-pass":
            if fmt != "packed":
                raise SystemExit("--lens two-pass needs --format packed")
            self.lens_map, _ = self.ctx.lens_map(self.lens_params, rows, cols)
        self.obs_fmt = fmt.upper() if fmt in ("uyvy", "yuy2", "i422", "i444", "ayuv") else None      # any other OBS format: lvk_hip_stab_push_obs
        self.yuv420 = fmt != "packed" and self.obs_fmt is None
        self.nv12 = fmt == "nv12"
        self.pingpong = pingpong
        self.file = None
        t_gen = time.perf_counter()
        if input_path is not None:
            # a clip from a file: raw I420 (Y, U, V planes back to back) or NV12 (Y, interleaved UV) frames, uploaded once
            if not self.yuv420:
                raise SystemExit("--input holds 4:2:0 frames: use --format i420 or nv12")
            fbytes = rows * cols * 3 // 2
            nfile = os.path.getsize(input_path) // fbytes
            if nfile < 2 * self.delay + 4:
                raise SystemExit(f"--input: {nfile} frames of {cols}x{rows} in {input_path}, need at least {2 * self.delay + 4}")
            pool = min(pool, nfile)
            self.file = np.memmap(input_path, dtype=np.uint8, mode="r", shape=(nfile, fbytes))
            self.clip = None
            planes = [tuple(torch.from_numpy(np.ascontiguousarray(q)).to(device) for q in self.file_planes(i)) for i in range(pool)]
        else:
            pool = max(pool, 2 * self.delay + 4)
            # ---- the synthetic stream, rendered on the GPU and kept resident in HBM (4K I420: 12.4 MB per frame, 7.5 GB for 600 poses)
            self.clip = clipgen.Clip(rows, cols, pool, seed=seed, device=device, cut_at=(pool // 2) if cut else None)
            planes = [self.clip.render_i420(i, nv12=self.nv12) for i in range(pool)] if self.yuv420 else None
        self.pool = pool
        if self.yuv420:
            self.planes = planes
            self.frames = None
            self.outs = [tuple(torch.empty_like(q) for q in planes[0]) for _ in range(4)]
            self.planes_args = [self.filt.prepare_yuv420(q) for q in planes]          # addresses / pitches marshalled once, outside the timed region
            self.outs_args = [self.filt.prepare_yuv420(o) for o in self.outs]
        elif self.obs_fmt is not None:
            # the source as an OBS source of that format delivers it: the generator's 4:4:4 frames through FrameIngest::to_obs, on the GPU
            def shapes():
                return {"UYVY": [(rows, cols, 2)], "YUY2": [(rows, cols, 2)], "AYUV": [(rows, cols, 4)], "I444": [(rows, cols)] * 3,
                        "I422": [(rows, cols), (rows, cols // 2), (rows, cols // 2)]}[self.obs_fmt]
            self.frames = None
            self.obs_planes = []
            for i in range(pool):
                pl = [torch.empty(sh, dtype=torch.uint8, device=device) for sh in shapes()]
                self.ctx.egress_obs(self.obs_fmt, self.clip.render444(i), pl)
                self.obs_planes.append(pl)
            self.ctx.sync()
            self.outs = [[torch.empty(sh, dtype=torch.uint8, device=device) for sh in shapes()] for _ in range(4)]
            self.planes_args = [self.filt.prepare_obs(self.obs_fmt, q) for q in self.obs_planes]
            self.outs_args = [self.filt.prepare_obs(self.obs_fmt, o) for o in self.outs]
        else:
            self.frames = [self.clip.render444(i) for i in range(pool)]
            self.outs = [torch.empty_like(self.frames[0]) for _ in range(4)]
        if self.lens_map is not None:
            self.lens_bufs = [torch.empty_like(self.frames[0]) for _ in range(self.delay + 4)]       # corrected frames stay borrowed for `delay` pushes
        torch.cuda.synchronize()
        self.t_gen = time.perf_counter() - t_gen
        self.step_no = 0

    def file_planes(self, i):
        """numpy planes of frame i of the --input file."""
        rows, cols = self.rows, self.cols
        f = self.file[i % self.file.shape[0]]
        y = f[:rows * cols].reshape(rows, cols)
        if self.nv12:
            return y, f[rows * cols:].reshape(rows // 2, cols // 2, 2)
        q = rows * cols // 4
        return y, f[rows * cols:rows * cols + q].reshape(rows // 2, cols // 2), f[rows * cols + q:].reshape(rows // 2, cols // 2)

    def index(self, i):
        if not self.pingpong:
            return i % self.pool
        period = 2 * self.pool - 2                           # forward, then backward: continuous motion without a wrap-around jump
        k = i % period
        return k if k < self.pool else period - k

    announce_always = bool(os.environ.get("LVK_BENCH_ANNOUNCE"))          # experiments: every push of every region announces its successor

    def step(self, announce=False):
        announce = announce or self.announce_always
        i = self.step_no; self.step_no += 1
        k = self.index(i)
        if self.yuv420:
            if announce:                # the next frame is resident already: lvk_hip_stab_prefetch_yuv420, then the push of this one
                self.filt.prefetch_yuv420_prepared(self.planes_args[self.index(i + 1)])
            return self.filt.apply_yuv420_prepared(self.planes_args[k], i, self.outs_args[i & 3])
        if self.obs_fmt is not None:
            return self.filt.apply_obs_prepared(self.planes_args[k], i, self.outs_args[i & 3])
        if self.lens_map is not None:
            corrected = self.ctx.remap_map(self.frames[k], self.lens_map, bg=(0, 0, 0), out=self.lens_bufs[i % len(self.lens_bufs)])     # LCFilter::filter
            return self.filt.apply(corrected, timestamp=i, out=self.outs[i & 3])
        return self.filt.apply(self.frames[k], timestamp=i, out=self.outs[i & 3])

    def sync(self):
        self.ctx.sync()                      # lvk_hip_sync: every stream of the filter (tracking, bulk, transfers)

    def close(self):
        self.filt.close(); self.ctx.close()


def run_region(rigs, n, device_sync, local_rank):
    """n free-running pushes on every rig (one host thread per rig beyond the first), bracketed by device-wide synchronisations.
    Returns (seconds, frames emitted by all rigs, host time stamps of rig 0's pushes)."""
    import gc
    import threading
    import torch
    if len(rigs) == 1:
        rig = rigs[0]
        # (as timeit does, and as the latency pass below does: the interpreter's collector is the host process, not the library under test -- a
        #  generation-0 pass showed as one 0.21 ms push in every ~15 of the 0.10 ms ones, 4 % of a --steps 20 region)
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            t0 = time.perf_counter()
            emitted = 0
            stamps = [t0]
            for _ in range(n):
                out, _ = rig.step()
                emitted += 1 if out is not None else 0
                stamps.append(time.perf_counter())
            device_sync()
            return time.perf_counter() - t0, emitted, stamps
        finally:
            if gc_was_on:
                gc.enable()
    gate = threading.Barrier(len(rigs) + 1)
    res = [None] * len(rigs)

    def worker(k):
        torch.cuda.set_device(local_rank)              # the current device is per host thread
        rig = rigs[k]
        gate.wait()
        e, st = 0, [time.perf_counter()]
        try:
            for _ in range(n):
                out, _ = rig.step()
                e += 1 if out is not None else 0
                st.append(time.perf_counter())
            res[k] = (e, st, None)
        except Exception as ex:                        # reported by the main thread
            res[k] = (e, st, ex)
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(rigs))]
    for t in threads:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    device_sync()
    elapsed = time.perf_counter() - t0
    for r in res:
        if r[2] is not None:
            raise r[2]
    return elapsed, sum(r[0] for r in res), res[0][1]


def latency_pass(rigs, n, local_rank):
    """n synchronised pushes (sync, push, sync) per rig, all rigs at once: per-rig lists of milliseconds."""
    def one(rig, out):
        for _ in range(n):
            rig.sync()
            t = time.perf_counter()
            rig.step()
            rig.sync()
            out.append((time.perf_counter() - t) * 1e3)
    lats = [[] for _ in rigs]
    # (as timeit does: the interpreter's collector must not stop every host thread in the middle of the pass -- with `max` in the line, one
    #  4-10 ms sample in all K streams at once was seen on two boxes of five; that is the host process, not the library under test)
    import gc
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        return _latency_pass(rigs, one, lats, local_rank)
    finally:
        if gc_was_on:
            gc.enable()


def _latency_pass(rigs, one, lats, local_rank):
    import threading
    import torch
    if len(rigs) == 1:
        one(rigs[0], lats[0])
        return lats

    def worker(k):
        torch.cuda.set_device(local_rank)
        one(rigs[k], lats[k])
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(rigs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return lats


def config_leg(lvk, local_rank, device, rows, cols, preset, lens, label, seed, steps=400, barrier=None, fmt="i420"):
    """A short leg of another BASELINE configuration (outside `value`): same generator, 48 poses played forward and backward, I420 planes
    resident in HBM, overlap on; free-running rate over `steps` pushes, p50 / p99 of 150 synchronised pushes, live remap time."""
    import torch
    rig = Rig(lvk, local_rank, device, seed, rows, cols, preset, fmt, lens, True, 48, cut=False, pingpong=True)
    try:
        for _ in range(rig.delay + 2 + 60):
            rig.step()
        rig.filt.set_profiling(True, stages=("remap",), every=4)
        rig.sync(); torch.cuda.synchronize()
        rig.filt.schedule_counters(reset=True)
        if barrier:
            barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            rig.step()
        rig.sync()
        dt = time.perf_counter() - t0
        sched = rig.filt.schedule_counters(reset=True)
        prof = rig.filt.profile()
        rig.filt.set_profiling(False)
        if barrier:
            barrier()
        lat = latency_pass([rig], 150, local_rank)[0]
        st = rig.filt.stats()
        rms, rn = prof["remap"]
        return {"workload": label, "value": steps / dt, "unit": "frames/s", "steps": steps, "frames": steps, "elapsed_s": dt, "p50_ms": float(np.percentile(lat, 50)),
                "p99_ms": float(np.percentile(lat, 99)), "remap_us": (rms / rn * 1e3) if rn else None, "trust": float(st.trust), "features": int(st.n_tracked),
                "schedule": sched}
    finally:
        rig.close()


def host_fed_leg(rig, nsteps, nlat, hpool=48, barrier=None):
    """SURVEY 8d's metric for host-resident frames: the rig's stream with the I420 planes in pinned HOST memory -- H2D of frame i + 1 on an
    upload stream while frame i is processed (one frame of look-ahead, lvk_hip_stab_prefetch_yuv420_host), the output planes written by the
    remap kernel itself into pinned host planes, no host synchronisation inside the loop.  Then `nlat` pushes one frame at a time (push +
    lvk_hip_sync: upload, track, remap, planes back).  barrier: called right before the timed loop and before the latency loop (N > 1: all
    ranks start together).  Also says on which NUMA node the pinned planes landed."""
    import torch
    from livevisionkit_amd import shard
    filt, ctx, rows, cols, nv12 = rig.filt, rig.ctx, rig.rows, rig.cols, rig.nv12
    hpool = min(rig.pool, hpool)
    host_in = [filt.host_planes(rows, cols, nv12) for _ in range(hpool)]            # pinned, contiguous I420 / NV12 frames (the OBS layout)
    for k in range(hpool):
        for dst, p in zip(host_in[k], rig.planes[k]):
            dst[...] = p.cpu().numpy()
    host_out = [filt.host_planes(rows, cols, nv12) for _ in range(4)]
    in_args = [filt.prepare_yuv420_host(p) for p in host_in]
    out_args = [filt.prepare_yuv420_host(p) for p in host_out]
    planes_node = shard.numa_node_of_address(host_in[0][0].ctypes.data)

    def run(n, base):
        # a streaming caller with frames queued ahead (VideoFilter::stream's reader thread): frame i + 1 is announced -- its upload
        # starts -- before frame i is pushed
        filt.prefetch_yuv420_host_prepared(in_args[base % hpool])
        for i in range(base, base + n):
            if i + 1 < base + n:
                filt.prefetch_yuv420_host_prepared(in_args[(i + 1) % hpool])
            filt.apply_yuv420_host_prepared(in_args[i % hpool], i, out_args[i & 3])
    torch.cuda.synchronize(); ctx.sync()
    run(100, rig.step_no); rig.step_no += 100
    ctx.sync()
    filt.schedule_counters(reset=True)
    if barrier:
        barrier()
    tp = time.perf_counter()
    run(nsteps, rig.step_no); rig.step_no += nsteps
    ctx.sync()
    dtp = time.perf_counter() - tp
    sched = filt.schedule_counters(reset=True)
    mb = rows * cols * 1.5 / 1e6
    # per-frame latency with the transfers inside (BASELINE's p99 ms/frame for host-resident frames): push the host planes, wait
    # for the emitted host planes -- one frame at a time
    if barrier:
        barrier()
    lat_pcie = []
    for i in range(rig.step_no, rig.step_no + nlat):
        tl = time.perf_counter()
        filt.apply_yuv420_host_prepared(in_args[i % hpool], i, out_args[i & 3])
        ctx.sync()
        lat_pcie.append((time.perf_counter() - tl) * 1e3)
    rig.step_no += nlat
    del host_in, host_out
    return {"value": nsteps / dtp, "unit": "frames/s", "frames": nsteps, "elapsed_s": dtp, "host_to_device_MB_per_frame": mb, "device_to_host_MB_per_frame": mb,
            "GBps_each_way": nsteps / dtp * mb / 1e3, "pinned_planes_numa_node": planes_node, "schedule": sched,
            "link_ceiling": "profiles/r03_pcie_probe.txt: 55 GB/s one way; both ways at once 46.8 GB/s each with a copy engine per direction (3760 frames/s), "
                            "43 with a copy engine up and kernel stores down (3470 frames/s -- what this path runs: the runtime performs D2H copies of this "
                            "process with a blit kernel), 36-40 with kernels both ways",
            "latency_ms": dict(percentiles(lat_pcie), samples=len(lat_pcie), note="lvk_hip_stab_push_yuv420_host + lvk_hip_sync per frame: upload, "
                               "track, remap written straight into the pinned output planes; one frame at a time"),
            "note": "lvk_hip_stab_push_yuv420_host with one frame of upload look-ahead (lvk_hip_stab_prefetch_yuv420_host): I420 planes in pinned host memory "
                    "in and out (SURVEY 8d's metric for host-resident frames), output planes written by the remap kernel itself, free-running; not the "
                    "headline value (inputs of `value` are resident in HBM)"}


def gpu_link_info(local_rank):
    """Where this rank's GPU hangs: NUMA node and PCIe link (sysfs current / max speed and width) -- with per-rank host-fed GB/s beside it,
    a slow rank of an 8-GPU run can be told from a narrow link or a remote socket."""
    import torch
    out = {"numa_node": -1, "pcie": None}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        d = os.path.join("/sys/bus/pci/devices", bdf)
        out["pci"] = bdf

        def rd(name):
            try:
                return open(os.path.join(d, name)).read().strip()
            except OSError:
                return None
        nn = rd("numa_node")
        out["numa_node"] = int(nn) if nn is not None else -1
        out["pcie"] = {"current_link_speed": rd("current_link_speed"), "current_link_width": rd("current_link_width"),
                       "max_link_speed": rd("max_link_speed"), "max_link_width": rd("max_link_width")}
    except Exception as e:
        out["error"] = repr(e)
    return out


def shared_resource_legs(lvk, rig, rank, local_rank, world, device, seed, barrier, host_fed=True, configs=True):
    """N > 1 (every rank, all at the same time, each leg between two barriers): the legs that CAN fail to scale -- N device-resident replicas share
    nothing, N host-fed streams share host DRAM, root ports and the inter-socket links (the metric includes H2D / D2H "when frames
    are host-resident"; 8e: one stream per device) -- and short legs of BASELINE configs 4 (1080p) and 5 (4K, lens pre-warp fused), which the
    headline's 4K plain stream does not cover.  Returns this rank's report; rank 0 aggregates (sum of the frames / max over ranks of the time)."""
    rep = {"rank": rank, "device": local_rank}
    rep.update(gpu_link_info(local_rank))
    legs = []
    if host_fed and rig.yuv420:
        legs.append(("host_fed", lambda b: host_fed_leg(rig, 600, 200, barrier=b)))
    if configs:
        for key, (r_, c_, lens_, label_) in (("config4_1080p", (1080, 1920, "off", "1920x1080 I420, OBS 'homography' preset (BASELINE config 4: one such stream per GPU)")),
                                             ("config5_4k_lens", (2160, 3840, "fused", "3840x2160 I420, lens pre-warp fused into the remap (BASELINE config 5: one such stream per GPU)"))):
            legs.append((key, lambda b, r_=r_, c_=c_, lens_=lens_, label_=label_: config_leg(lvk, local_rank, device, r_, c_, "homography", lens_, label_, seed + 101, steps=400, barrier=b)))
    # (a leg that fails half-way on this rank still meets the other ranks at its two barriers: lvk.shard.run_legs)
    rep.update(lvk.shard.run_legs(legs, barrier, per_leg=2))
    return rep


def aggregate_legs(reports):
    """Whole-job figures of shared_resource_legs over the ranks: frames of all ranks / the slowest rank's time, per-rank GB/s and p50 / p99."""
    out = {"ranks": reports}
    for key in ("host_fed", "config4_1080p", "config5_4k_lens"):
        legs = [r.get(key) for r in reports]
        if not legs or any(l is None or "error" in l for l in legs):
            out[key] = None if not any(legs) else {"error": [l.get("error") if l else "missing" for l in legs]}
            continue
        frames = sum(l.get("frames", l.get("steps", 0)) for l in legs)
        slowest = max(l["elapsed_s"] for l in legs)
        agg = {"value": frames / slowest, "unit": "frames/s", "frames": frames, "slowest_rank_s": slowest,
               "per_rank_frames_per_s": [l["value"] for l in legs]}
        if key == "host_fed":
            agg["per_rank_GBps_each_way"] = [l["GBps_each_way"] for l in legs]
            agg["per_rank_p50_ms"] = [l["latency_ms"]["p50"] for l in legs]; agg["per_rank_p99_ms"] = [l["latency_ms"]["p99"] for l in legs]
            agg["per_rank_planes_numa_node"] = [l.get("pinned_planes_numa_node") for l in legs]
        else:
            agg["per_rank_p50_ms"] = [l["p50_ms"] for l in legs]; agg["per_rank_p99_ms"] = [l["p99_ms"] for l in legs]
        agg["p99_ms"] = max(agg["per_rank_p99_ms"]); agg["p50_ms"] = max(agg["per_rank_p50_ms"])
        out[key] = agg
    return out


def multi_stream_leg(lvk, local_rank, device, seed, rows, cols, preset, fmt, lens, overlap, Km):
    """Km independent filters (own HIP streams, own clips, one host thread each) on this ONE GPU: free-running aggregate over 600 pushes per
    stream, then every stream pushing synchronised frames at once (per-stream p50 / p99)."""
    try:
        mr = [Rig(lvk, local_rank, device, seed + k, rows, cols, preset, fmt, lens, overlap, 48, cut=False, pingpong=True) for k in range(Km)]
        try:
            for r in mr:
                for _ in range(r.delay + 2):
                    r.step()
            run_region(mr, 100, lambda: None, local_rank)
            for r in mr:
                r.sync()
            dtm, em, _ = run_region(mr, 600, lambda: [r.sync() for r in mr], local_rank)
            lm = latency_pass(mr, 150, local_rank)
            return {"streams": Km, "preset": preset, "value": em / dtm, "unit": "frames/s", "pushes_per_stream": 600,
                    "per_stream_latency_ms": [dict(percentiles(x), samples=len(x)) for x in lm],
                    "trust": [float(r.filt.stats().trust) for r in mr],
                    "note": f"{Km} independent filters (own HIP streams, own clips, one host thread each) on this ONE GPU, OBS '{preset}' preset, free-running "
                            "aggregate, then every stream pushing synchronised frames at once; `bench.py --streams-per-gpu K [--preset field]` makes it the headline"}
        finally:
            for r in mr:
                r.close()
    except Exception as e:          # an extra leg must never break the contract line
        return {"error": repr(e)}


def reference_kernel_leg(rig):
    """The REFERENCE's own remap kernel -- FSR.cl's easu_remap_homography compiled for gfx950 from the reference tree (oracle/_ref/fsr_yuv.hsaco,
    recipe oracle/Makefile `ref`) and launched the way lvk::remap launches it (Functions/Image.cpp:133-146, 8 x 8 work-groups of
    OpenCL/Kernels.cpp:49-71) -- timed with HIP events on one frame of the workload with the warp that stabilizes it, next to the product's
    kernel on the same frame, same matrix, same stream; and whether the two outputs are identical."""
    import torch
    from tests import ref_cl
    if not ref_cl.available():
        return {"error": "oracle/_ref/*.hsaco not present (built by __graft_entry__.build() where /root/reference exists)"}
    ref = ref_cl.RefKernels()
    rows, cols = rig.rows, rig.cols
    if rig.clip is not None:
        src = rig.clip.render444(7)
        H = np.linalg.inv(rig.clip.matrix(7)) @ rig.clip.matrix(7, smooth=True)          # stabilized pixel -> source pixel of frame 7
    else:
        src = rig.ctx.ingest_yuv420(*rig.planes[7])
        th, z = np.deg2rad(0.15), 1.002
        c, si = np.cos(th) * z, np.sin(th) * z
        H = np.array([[c, -si, 0.004 * cols], [si, c, -0.003 * cols], [0, 0, 1.0]])
    H = H / H[2, 2]
    bg = (105, 212, 235)
    out_r = torch.zeros_like(src); out_g = torch.zeros_like(src)
    torch.cuda.synchronize(); rig.sync()

    def timed(fn, n):
        with torch.cuda.stream(rig.stream):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(rig.stream)
            for _ in range(n):
                fn()
            e1.record(rig.stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    # both launches with their arguments marshalled once: the loops below enqueue faster than the kernels run, so the event pair brackets
    # GPU time (a Python-side argument build per launch is ~0.1-2 ms and would be what the events measure)
    go_ref, _ = ref.remap_homography(src, H, bg=bg, yuv=True, out=out_r, prepared=True)
    lib, hnd = rig.ctx.lib, rig.ctx.handle
    import ctypes
    Hf = (ctypes.c_float * 9)(*[float(np.float32(v)) for v in H.reshape(-1)])
    bgc = (ctypes.c_uint8 * 3)(*bg)
    a_hip = (hnd, ctypes.c_void_p(src.data_ptr()), src.stride(0), rows, cols, ctypes.c_void_p(out_g.data_ptr()), out_g.stride(0), rows, cols, 0, 0, Hf, bgc, 1)

    def go_hip():
        rc = lib.lvk_hip_remap_homography(*a_hip)
        assert rc == 0, rc
    t_ref = timed(go_ref, 20)
    t_hip = timed(go_hip, 40)
    same = bool(torch.equal(out_r, out_g))
    b = 6 * rows * cols
    return {"kernel": "easu_remap_homography (LiveVisionKit/Functions/OpenCL/Sources/FSR.cl, -D YUV_INPUT) compiled for gfx950 by oracle/Makefile `ref`",
            "launch": "Functions/Image.cpp:133-146 argument list, 8x8 work-groups (OpenCL/Kernels.cpp:49-71)", "frame": f"{cols}x{rows} packed YUV444 in / out ({b} B)",
            "avg_launch_us": t_ref, "launches": 20, "hbm_frac": b / (t_ref * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "product_kernel": "k_remap_homography<yuv> (same frame, same matrix, same stream, alone on the GPU)", "product_avg_launch_us": t_hip,
            "product_hbm_frac": b / (t_hip * 1e-6) / 1e9 / HBM_PEAK_GBS, "speedup": t_ref / t_hip, "outputs_bit_equal": same}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # never a silent degradation: the ranks that run are the ranks that were asked for
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    K = max(1, args.streams_per_gpu)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    share = os.environ.get("LVK_BENCH_SHARE_GPU") == "1"       # functional test of the N > 1 path on a box with fewer GPUs than ranks
    if world > torch.cuda.device_count() and not share:
        raise SystemExit(f"bench.py: --gpus {world} but {torch.cuda.device_count()} GPU(s) visible (LVK_BENCH_SHARE_GPU=1 maps the ranks onto them for a functional test)")
    if share:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")              # timing barrier / reductions only: the data path has no collective

    import livevisionkit_amd as lvk
    from tests import clipgen
    numa_cpus = lvk.shard.bind_to_gpu_numa(local_rank) if os.environ.get("LVK_BENCH_NUMA", "1") != "0" else []
    rows, cols = args.rows, args.cols
    seed0 = 0x4C564B31 + rank * K
    pool_k = args.pool if K == 1 else max(48, min(args.pool, 600 // K))      # K clips share the HBM budget of one
    rigs = [Rig(lvk, local_rank, device, seed0 + k, rows, cols, args.preset, args.format, args.lens, not args.no_overlap,
                pool_k, input_path=args.input) for k in range(K)]
    rig = rigs[0]
    filt, ctx, clip, delay, pool = rig.filt, rig.ctx, rig.clip, rig.delay, rig.pool
    settings, lens_params = rig.settings, rig.lens_params
    yuv420, nv12 = rig.yuv420, rig.nv12
    t_gen = sum(r.t_gen for r in rigs)
    if args.write_input:
        if not yuv420 or clip is None:
            raise SystemExit("--write-input writes the synthetic clip's 4:2:0 planes")
        with open(args.write_input, "wb") as f:
            for q in rig.planes:
                for plane in q:
                    f.write(plane.cpu().numpy().tobytes())
        print(json.dumps({"written": args.write_input, "frames": pool, "rows": rows, "cols": cols, "format": args.format}), flush=True)
        return None

    def step():
        return rig.step()

    # the sensor sampler (a separate process) is started BEFORE the warmup, so that nothing but the barrier sits between the warmup steps
    # and the timed region (its start-up wait used to idle the GPU for 0.3 s right in front of the timed pushes)
    sampler = None
    if rank == 0 and not args.no_sensors:
        try:
            props = torch.cuda.get_device_properties(local_rank)
            bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id) if hasattr(props, "pci_bus_id") else None
            sampler = GpuSampler(bus)
            time.sleep(0.3)                                  # the sampler process is up before the timed region begins
        except Exception:
            sampler = None
    # fill the pipeline (untimed, before the --warmup steps): the frame delay, so that every timed step emits one stabilized frame; the
    # detector's start-up burst -- a new stream runs FAST on most of its first 2 N pushes while the suppression grid fills up (pushes of 0.13-0.15 ms
    # instead of 0.10); and the chip's clock / power ramp -- 35 pushes are 4 ms of GPU work after seconds of rendering the clip at a low duty cycle
    # (measured: pushes of 0.11-0.135 ms there against 0.100-0.105 ms a few hundred pushes later, same kernels).  None of the three is the steady state
    # SURVEY 8d's metric is defined on ("after the N-frame delay has filled, >= 600 frames"): 1000 pushes (>= 2 N + 10; ~0.1 s at 4K -- a fixed
    # count, so that two runs of the same command push the same frames), whatever --warmup is, so that the driver's `--steps 20 --warmup 5` region
    # samples the same stream the default run does.
    n_fill = max(delay + 2, 2 * delay + 10, 1000)
    if K == 1:
        for _ in range(n_fill):
            step()
    else:
        run_region(rigs, n_fill, lambda: None, local_rank)
    fps_ = max(1, args.frames_per_step)
    n_timed, n_warm = args.steps * fps_, args.warmup * fps_
    if K == 1:
        for _ in range(n_warm):
            step()
    else:
        run_region(rigs, n_warm, lambda: None, local_rank)
    # live HIP-event timing of the dominant kernel inside the timed region: that stage only, and one launch in eight (an event pair per
    # frame is two host API calls in the per-frame turnaround -- the measurement would slow what it measures by ~4 %)
    filt.set_profiling(True, stages=("remap",), every=8)

    def device_sync():
        for r in rigs:
            r.sync()
        torch.cuda.synchronize()

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    barrier()
    schedule = {}
    filt.schedule_counters(reset=True)
    wall0 = time.time()
    elapsed, emitted, stamps = run_region(rigs, n_timed, device_sync, local_rank)
    wall1 = time.time()
    schedule["timed_region"] = filt.schedule_counters(reset=True)
    free_running = np.diff(np.array(stamps)) * 1e3          # host time per push of stream 0 in the free-running timed region
    barrier()
    # SUSTAINED rate ("steady state, >= 600 frames"): the same free-running loop kept going for at least 600 more
    # pushes right behind the timed region -- a --steps 20 timed region is 2.6 ms, of which the pipeline fill after the barrier and the
    # un-overlapped last remap are 6-8 %.  Outside `value` (the contract times exactly --steps), reported beside it; it also gives the
    # roofline its >= 64 event samples of the remap whatever --steps was.
    n_sustained = max(600, 64 * 8 - n_timed)
    wall2 = time.time()
    sustained_s, sustained_emitted, _ = run_region(rigs, n_sustained, device_sync, local_rank)
    wall3 = time.time()
    schedule["sustained"] = filt.schedule_counters(reset=True)
    sensor_samples = sampler.stop() if sampler is not None else []
    prof = filt.profile()
    filt.set_profiling(False)
    stats = filt.stats()
    import zlib
    last_out = rig.outs[(rig.step_no - 1) & 3]
    out_crc = 0
    for q in (last_out if isinstance(last_out, tuple) else (last_out,)):
        out_crc = zlib.crc32(q.cpu().numpy().tobytes(), out_crc)

    # per-step latency pass (each step synchronised) for p50 / p99 ms per frame: always 500 pushes, whatever --steps was
    # (K > 1: every stream at once, one host thread each -- the latency a stream sees next to its K - 1 neighbours)
    lats = latency_pass(rigs, 500 if K == 1 else 200, local_rank)
    lat = lats[0]
    schedule["latency_pass"] = filt.schedule_counters(reset=True)
    # ... and the same push the way a LIVE source makes it (the plugin's pattern: one frame every 16.7 ms, the GPU idle in between): 90 frames paced at
    # 60 fps, push + sync each -- whether clock / power management between frames costs a 4K60 stream latency (scripts/idle_probe.py sweeps the gap)
    paced = None
    if rank == 0 and K == 1:
        paced_ms = []
        for _ in range(90):
            rig.sync()
            t_next = time.perf_counter() + 1.0 / 60.0
            while time.perf_counter() < t_next:
                pass
            t_ = time.perf_counter(); rig.step(); rig.sync(); paced_ms.append((time.perf_counter() - t_) * 1e3)
        paced = dict(percentiles(paced_ms), samples=len(paced_ms), note="one synchronised push every 1 / 60 s (GPU idle for 16.5 ms before each)")
        filt.schedule_counters(reset=True)

    # every stage's event timing in a separate free-running pass (outside the timed region: 16 event records per frame)
    filt.set_profiling(True)
    for _ in range(300):
        step()
    device_sync()
    prof_all = filt.profile()
    filt.set_profiling(False)

    # N > 1: the legs that can fail to scale, on every rank at once (host-fed frames; BASELINE configs 4 and 5)
    leg_reports = None
    if world > 1 and K == 1 and not args.no_overlap:
        mine = shared_resource_legs(lvk, rig, rank, local_rank, world, device, seed0, barrier, host_fed=not args.no_pcie, configs=not args.no_configs and args.input is None)
        leg_reports = lvk.shard.gather_rank_reports(mine)

    # the same stream by a caller that knows its next frame (VideoFilter::stream's reader thread is one ahead; a transcoder): frame i + 1 is
    # announced before frame i is pushed (lvk_hip_stab_prefetch_yuv420), its downscale + pyramid run behind frame i's chain.  Beside `value`.
    lookahead = None
    if rank == 0 and world == 1 and K == 1 and yuv420 and not args.no_lookahead:
        try:
            device_sync()
            for _ in range(100):
                rig.step(announce=True)
            device_sync()
            hits0 = filt.lookahead_frames()
            tl = time.perf_counter()
            for _ in range(1000):
                rig.step(announce=True)
            device_sync()
            dtl = time.perf_counter() - tl
            hits = filt.lookahead_frames() - hits0
            lat_la = []
            for _ in range(300):
                rig.sync(); t_ = time.perf_counter(); rig.step(announce=True); rig.sync(); lat_la.append((time.perf_counter() - t_) * 1e3)
            filt.prefetch_cancel()
            lookahead = {"frames_per_s": 1000 / dtl, "frames": 1000, "pushes_that_found_their_pyramid_built": int(hits),
                         "latency_ms": dict(percentiles(lat_la), samples=len(lat_la)),
                         "note": "free-running, frame i + 1 announced (lvk_hip_stab_prefetch_yuv420) before frame i is pushed; same pixels "
                                 "(tests/test_stabilizer_gpu.py::test_device_lookahead_same_frames); not `value`, whose caller knows one frame at a time"}
        except Exception as e:          # the extra pass must never break the contract line
            lookahead = {"error": repr(e)}

    # the same remap kernel alone on the GPU at full occupancy (the timed region runs its occupancy-capped `_co` variant next to the tracker)
    standalone_us = None
    if rank == 0 and world == 1 and K == 1 and args.lens != "two-pass" and clip is not None:
        device_sync()
        meshes = filt.meshes()[1]
        srcs = rig.frames[:8] if rig.frames is not None else [clip.render444(i) for i in range(8)]
        dst = torch.empty_like(srcs[0])
        torch.cuda.synchronize()                                 # the frames are rendered on torch's stream, the remap runs on the filter's
        bgc = tuple(int(v) for v in settings.background)
        with torch.cuda.stream(rig.stream):
            for _ in range(3):
                ctx.warpmesh_apply(srcs[0], meshes, bg=bgc, yuv=True, out=dst)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(rig.stream)
            for i in range(40):
                ctx.warpmesh_apply(srcs[i % len(srcs)], meshes, bg=bgc, yuv=True, out=dst)
            e1.record(rig.stream)
        torch.cuda.synchronize()
        standalone_us = e0.elapsed_time(e1) / 40 * 1e3
        del srcs, dst

    # PCIe-inclusive rate (reported beside `value`, never as it): the same stream with the I420 planes in pinned HOST memory (host_fed_leg).
    # world == 1: rank 0's own leg, here; world > 1: EVERY rank runs it at the same time, barrier-aligned (shared_resource_legs below) -- host
    # DRAM, the root ports and the inter-socket links are what N device-resident replicas do not share and N host-fed streams do.
    pcie = None
    if rank == 0 and world == 1 and K == 1 and yuv420 and not args.no_pcie and not args.no_overlap:
        try:
            pcie = host_fed_leg(rig, 1000, 500)
        except Exception as e:          # the extra pass must never break the contract line
            pcie = {"error": repr(e)}

    # ---- beside the headline (rank 0 of a 1-GPU run, outside `value`): the reference's own kernel on this chip, the other BASELINE
    # configurations, K streams on this GPU
    extras = rank == 0 and world == 1 and K == 1
    reference_kernel = None
    if extras and not args.no_reference_kernel:
        try:
            reference_kernel = reference_kernel_leg(rig)
        except Exception as e:
            reference_kernel = {"error": repr(e)}
    configs = None
    if extras and not args.no_configs and args.input is None:
        configs = []
        for (r_, c_, preset_, lens_, label_) in (
                (1080, 1920, "homography", "off", "1920x1080 I420, OBS 'homography' preset (BASELINE configs 2 / 4, per-GPU leg)"),
                (2160, 3840, "homography", "fused", "3840x2160 I420, lens correction fused into the remap (BASELINE config 5, per-GPU leg)"),
                (2160, 3840, "field", "off", "3840x2160 I420, OBS 'vector field' preset (16x16 mesh)")):
            try:
                configs.append(config_leg(lvk, local_rank, device, r_, c_, preset_, lens_, label_, seed0 + 101))
            except Exception as e:
                configs.append({"workload": label_, "error": repr(e)})
        # a source that is not 4:2:0 (what a capture card delivers): the plugin's whole path through lvk_hip_stab_push_obs -- P422Ingest::to_ocl, the
        # filter, ::to_obs (FrameIngest.cpp:604-666); its remap writes the UYVY pairs itself (csrc/remap_obs.hip)
        label_ = "1920x1080 UYVY (packed 4:2:2) through lvk_hip_stab_push_obs, OBS 'homography' preset"
        try:
            configs.append(config_leg(lvk, local_rank, device, 1080, 1920, "homography", "off", label_, seed0 + 102, fmt="uyvy"))
        except Exception as e:
            configs.append({"workload": label_, "error": repr(e)})
    multi_stream = multi_stream_field = None
    if extras and not args.no_multi_stream and args.input is None and yuv420:
        multi_stream = multi_stream_leg(lvk, local_rank, device, seed0 + 200, rows, cols, args.preset, args.format, args.lens, not args.no_overlap, 4)
        if args.preset != "field" and not args.no_configs:
            # the vector-field preset's motion stage is ~144 us of dependent pivots on 5 workgroups: what K concurrent field streams deliver
            # says whether that chain or the remap bounds a deployment
            multi_stream_field = multi_stream_leg(lvk, local_rank, device, seed0 + 300, rows, cols, "field", args.format, args.lens, not args.no_overlap, 4)

    elapsed_max, total_frames = lvk.shard.reduce_timing(elapsed, emitted)
    sustained_max, sustained_frames = lvk.shard.reduce_timing(sustained_s, sustained_emitted)
    rank_reports = lvk.shard.gather_rank_reports({
        "rank": rank, "device": local_rank, "clip_seed": seed0, "streams": K, "frames": emitted, "elapsed_s": elapsed, "frames_per_s": emitted / elapsed,
        "sustained_frames_per_s": sustained_emitted / sustained_s, "link": gpu_link_info(local_rank), "numa_cpus": (f"{numa_cpus[0]}-{numa_cpus[-1]} ({len(numa_cpus)})" if numa_cpus else "unbound"),
        # every rank reports the latency of every stream it ran (north star: throughput AND p99 at 1 / 2 / 4 / 8 GPUs): p50 / p99 per stream,
        # and the rank's own figure = its slowest stream
        "latency_ms": {"p50": max(float(np.percentile(x, 50)) for x in lats), "p99": max(float(np.percentile(x, 99)) for x in lats),
                       "max": max(float(np.max(x)) for x in lats), "samples": len(lats[0])},
        "stream_latency_ms": [dict(percentiles(x), samples=len(x)) for x in lats]})

    result = None
    rc = 0
    if rank == 0:
        remap_ms, remap_n = prof["remap"]
        remap_s = remap_ms / remap_n * 1e-3 if remap_n else None
        # S_in + S_out of the frames the dominant kernel reads / writes as stored on the device: the packed 8UC3
        # frame in; out = packed 8UC3 (6 W H in total) or, on the 4:2:0 path, the planes of the fused remap + egress kernel (4.5 W H)
        fused_420 = yuv420 and args.lens != "two-pass"
        alg_bytes = (9 * rows * cols) // 2 if fused_420 else 6 * rows * cols
        achieved = (alg_bytes / remap_s) / 1e9 if remap_s else 0.0
        # Counter-derived figures come from the committed rocprofv3 summaries of THIS command under profiles/ (separate --pmc runs, never measured
        # by the timed run itself): a key that is missing there is reported as null, never replaced by a constant (tests/test_bench_profiles.py
        # holds every key read here to the committed files).
        traffic = counters = stalls = None
        kernel_id = ("k_remap_homography" if args.preset == "homography" else "k_remap_mesh") + ("_lens" if args.lens == "fused" else "") + ("_420" if fused_420 else "")
        for fname in PROFILE_FILES["traffic"]:
            tpath = os.path.join(ROOT, "profiles", fname)
            if os.path.exists(tpath):
                try:
                    t = json.load(open(tpath))
                    if t.get("rows") == rows and t.get("cols") == cols and t.get("kernel", "").split("<")[0] == kernel_id:
                        traffic = t.get("hbm_bytes_per_launch")
                        counters = t
                except Exception:
                    pass
        spath = os.path.join(ROOT, "profiles", PROFILE_FILES["stalls"])
        if os.path.exists(spath):
            try:
                stalls = (json.load(open(spath)).get("live") or {}).get(kernel_id) if rows == 2160 and cols == 3840 else None
            except Exception:
                stalls = None
        valu_per_px = (counters or {}).get("valu_per_px")                  # rocprofv3 SQ_INSTS_VALU * 64 / pixels
        valu_per_px_packed = (counters or {}).get("valu_per_px_packed")
        valu_rate = valu_per_px * rows * cols / remap_s if (remap_s and valu_per_px) else None
        clock_mhz = (stalls or {}).get("clock_mhz_under_kernel")
        peak_at_clock = VALU_PEAK_SPEC * clock_mhz / SPEC_CLOCK_MHZ if clock_mhz else None
        stage_us = {k: (v[0] / v[1] * 1e3 if v[1] else 0.0) for k, v in prof_all.items()}
        # the HBM-bound kernels beside the remap (north star: "HBM GB/s on the remap and pyramid kernels against the chip's peak"): the luma
        # downscale that feeds the pyramid and the 4:2:0 -> 4:4:4 conversion, from the live HIP-event times of the all-stages pass
        secondary = []
        det = settings.detection_width * settings.detection_height
        for kname, stage, b, what in (
                ("k_area_fast_dw / k_area_general* (luma INTER_AREA downscale to the tracking resolution)", "downscale", rows * cols + det, "W H luma read + the tracking frame written"),
                ("k_pyr_fused3 (pyramid levels 1-3 in one launch)", "pyramid", det + det // 4 + det // 16 + det // 64, "level 0 read + levels 1-3 written (172 KB: launch latency, not bandwidth)"),
                ("k_ingest_yuv420_x2 (4:2:0 planes -> packed 4:4:4)", "ingest", (9 * rows * cols) // 2 if yuv420 else 0, "1.5 W H read + 3 W H written")):
            us = stage_us.get(stage, 0.0)
            if us > 0 and b > 0:
                secondary.append({"kernel": kname, "bound": "hbm", "algorithmic_bytes_per_launch": b, "bytes": what, "avg_launch_us": us,
                                  "achieved": b / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  # short kernels: an event pair brackets the kernel AND the two markers' own latency (~2.5-3.5 us); the rocprofv3
                                  # means of the same command are the kernels' own durations
                                  "timing": "HIP events on the launch stream, all-stages pass (a lower bound on `frac`: rocprofv3 means in "
                                            "profiles/r06_kernel_stats.csv are 5.9 us for the downscale = 18 %, 9.5 us for the conversion = 49 %)"})
        n_ranks = len(rank_reports)
        legs_agg = aggregate_legs(leg_reports) if leg_reports else None
        result = {
            "metric": ("stabilized frames/sec (one 4K YUV420 stream per GPU, steady state)" if K == 1 else f"stabilized frames/sec ({K} concurrent 4K YUV420 streams per GPU, steady state)") if yuv420 else
                      "stabilized frames/sec (one 4K packed-YUV444 stream per GPU, steady state)",
            "value": total_frames / elapsed_max,
            "unit": "frames/s",
            "n_gpus": n_ranks,                               # the ranks that timed
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,          # one step = frames_per_step consecutive frames (config.frames_per_step)
            "ms_per_frame": elapsed_max / n_timed * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.input is None else f"file {os.path.basename(args.input)} ({pool} frames resident in HBM, cycled)",
            "config": {"workload": (f"{cols}x{rows} {args.format.upper()} (YUV 4:2:0) stream per GPU, planes resident in HBM, ingest -> lvk::StabilizationFilter -> egress, "
                                    if yuv420 else f"{cols}x{rows} packed YUV444 8UC3 stream per GPU (lvk::StabilizationFilter boundary format), ")
                                   + f"OBS '{args.preset}' preset, tracking 480x270, predictive_samples={delay}, crop 5%"
                                   + ("" if args.lens == "off" else f", lens correction {args.lens} (fx=fy=0.8W, k1=-0.12, k2=0.03)")
                                   + ("" if K == 1 else f"; {K} such streams per GPU, one host thread each"),
                       "clip": (f"SURVEY 8d generator: {pool} distinct poses (smooth pan + AR(1) jitter: 0.4 % W translation, 0.15 deg, 0.2 % zoom), scene cut at frame {pool // 2}, "
                                f"cycled; rendered on the GPU in {t_gen:.1f} s") if args.input is None else f"{args.input}: {pool} frames, cycled",
                       "parallelism": f"{n_ranks} rank(s), one per GPU, {K} independent stream(s) each, no collective (gloo barrier only)",
                       "streams_per_gpu": K, "frames_in_hbm": pool * K, "host_cpus_bound": len(numa_cpus),
                       "frames_per_step": fps_,
                       "step": f"one step = one batch of {fps_} consecutive frames of the stream = {fps_} pushes (lvk_hip_stab_push_yuv420), each emitting one stabilized frame; "
                               f"the timed region is exactly {args.steps} steps = {n_timed} pushes per stream",
                       "pipeline_fill": f"{n_fill} untimed pushes (frame delay, the detector's start-up burst, ~0.1 s for the chip's clock ramp) before the {args.warmup} warmup steps",
                       # copies of the line's steady-state figures (>= 600 frames, p50 / p99 of a synchronised push) where a parser that keeps
                       # only the contract's keys still finds them
                       "steady_state": {"frames_per_s": sustained_frames / sustained_max, "frames": int(sustained_frames),
                                        "p50_ms": max(r["latency_ms"]["p50"] for r in rank_reports), "p99_ms": max(r["latency_ms"]["p99"] for r in rank_reports)},
                       "shared_resource_legs": ({k: ({kk: v[kk] for kk in ("value", "p50_ms", "p99_ms", "per_rank_frames_per_s") if kk in v} if isinstance(v, dict) else v)
                                                 for k, v in legs_agg.items() if k != "ranks"} if legs_agg else None)},
            # SURVEY 8d's metric as it is defined (steady state, >= 600 frames; p50 / p99 of one synchronised push) beside the contract's `value`
            "value_sustained": sustained_frames / sustained_max, "p50_ms": max(r["latency_ms"]["p50"] for r in rank_reports),
            "p99_ms": max(r["latency_ms"]["p99"] for r in rank_reports),
            "sustained": {"frames": int(sustained_frames), "frames_per_s": sustained_frames / sustained_max, "ms_per_frame": sustained_max / n_sustained * 1e3,
                          "note": "the free-running loop continued for >= 600 pushes right after the timed region (SURVEY 8d's steady state); "
                                  "whole job, max over ranks; not `value`"},
            "gpu_sensors": {"timed_region": GpuSampler.window(sensor_samples, wall0, wall1), "sustained_region": GpuSampler.window(sensor_samples, wall2, wall3),
                            "source": "amdgpu sysfs hwmon (freq1_input = shader clock, power1_average = socket power), sampled every 2 ms by a separate process"}
                           if sensor_samples else None,
            "ranks": rank_reports,
            # whole job: the slowest rank's (= slowest stream's) p50 / p99 of the synchronised pushes; every rank's own pair is in `ranks`
            "latency_ms": {"p50": max(r["latency_ms"]["p50"] for r in rank_reports), "p99": max(r["latency_ms"]["p99"] for r in rank_reports),
                           "max": max(r["latency_ms"]["max"] for r in rank_reports), "samples": len(lat), "over": f"max over {n_ranks} rank(s) x {K} stream(s); per rank / per stream: ranks[].latency_ms, ranks[].stream_latency_ms"},
            "latency_paced_60fps_ms": paced,
            "extras": ("single-rank only: pcie_inclusive, lookahead, reference_kernel, configs, multi_stream, cpu_baseline, quality and roofline.standalone_* run on "
                       "rank 0 of a --gpus 1 --streams-per-gpu 1 run and are null otherwise (N > 1: the host-fed leg and configs 4 / 5 run on EVERY rank at once, "
                       "multi_gpu_legs); free_running_ms / timed_region_ms / stage_us / tracking / roofline / schedule are rank 0's stream 0"),
            # N > 1 only: host-fed frames and BASELINE configs 4 / 5 on every rank at the same time -- aggregate, per-rank GB/s, p50 / p99, NUMA node, PCIe link
            "multi_gpu_legs": legs_agg,
            # which schedule the library chose for the pushes of each region (lvk_hip_stab_schedule_counters): free-running / persistent grid in the
            # timed regions, synchronised / full grid / signal word in the latency pass -- or a box on which it is not
            "schedule": schedule,
            "free_running_ms": percentiles(free_running, (10, 50, 90, 99)),
            # where the timed region goes: host time of each timed push (the first one has nothing to overlap with, pushes on which the
            # detector runs are longer) and what is left for the final synchronisation (the last remap + lvk_hip_sync)
            "timed_region_ms": {"pushes": [round(float(x), 4) for x in free_running[:64]],
                                "final_sync": round(float(elapsed * 1e3 - free_running.sum()), 4), "total": round(float(elapsed * 1e3), 4)},
            "stage_us": stage_us,
            "pcie_inclusive": pcie,
            "lookahead": lookahead,
            "tracking": {"stability": stats.tracking_stability, "trust": stats.trust, "features": stats.n_tracked, "last_output_crc32": out_crc},
            "roofline": {"kernel": ("k_remap_homography" if args.preset == "homography" else "k_remap_mesh") + ("_lens" if args.lens == "fused" else "")
                                   + (("_420<nv12>" if nv12 else "_420<i420>") if fused_420 else ("<yuv>" if args.no_overlap else "_co<yuv>")),
                         # The HBM figures are what the contract asks for (algorithmic bytes / launch duration against 8 TB/s).  The
                         # counters say the kernel is bound by fp32 VALU issue, not by HBM (traffic ~ algorithmic bytes, SQ_INSTS_VALU
                         # ~ the whole SIMD time): `binding` names that roofline and `valu_*` price the kernel against it.
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         # NOT measured by this run: rocprofv3 --pmc passes of the same command (scripts/pmc_remap.sh), kept under profiles/
                         "traffic_source": "profiles/remap_pmc_traffic*.json (separate rocprofv3 --pmc run of this command; FETCH_SIZE x2 + WRITE_SIZE per launch)" if traffic else None,
                         "valu_instr_per_px_source": "profiles/remap_pmc_traffic*.json (SQ_INSTS_VALU x 64 / pixels, separate rocprofv3 --pmc run)" if counters else None,
                         "avg_launch_us": remap_s * 1e6 if remap_s else None, "launches": remap_n,
                         "binding": "valu",
                         # the fraction of the roofline that BINDS this kernel (fp32 VALU issue, spec peak at 2.4 GHz): the line's first number to read
                         "binding_frac": valu_rate / VALU_PEAK_SPEC if valu_rate else None,
                         "valu_instr_per_px": valu_per_px,
                         "valu_achieved_Tlaneops": valu_rate / 1e12 if valu_rate else None,
                         "valu_peak_spec_Tlaneops": VALU_PEAK_SPEC / 1e12, "valu_frac_spec": valu_rate / VALU_PEAK_SPEC if valu_rate else None,
                         # the same rate against the issue peak at the clock the chip RUNS THIS KERNEL at (GRBM_GUI_ACTIVE / 8 XCDs / duration of the
                         # committed counter run, profiles/r06_remap_stalls.txt) -- never against a probe's clock
                         "clock_mhz_under_kernel": clock_mhz,
                         "valu_frac_at_kernel_clock": valu_rate / peak_at_clock if (valu_rate and peak_at_clock) else None,
                         # counters only (same committed run): VALU issue slots used = SQ_INSTS_VALU x 2 cycles / (cycles x 1024 SIMDs), and where the
                         # waves' time went (ACTIVE_INST_ANY + WAIT_INST_ANY [ready, not issued: issue arbitration] + WAIT_ANY [s_waitcnt: memory] = 1)
                         "valu_busy_frac": (stalls or {}).get("valu_issue_slot_frac"),
                         "wave_time": {k: (stalls or {}).get(k) for k in ("active_inst_any_frac", "wait_inst_any_frac", "wait_any_frac", "waves_per_simd")} if stalls else None,
                         "stall_counters_source": ("profiles/" + PROFILE_FILES["stalls"] + " (scripts/pmc_stalls.sh: SQ_* + GRBM_GUI_ACTIVE of the shipped kernel, the pipeline's own launches)") if stalls else None,
                         # the full-occupancy kernel alone on the GPU (outside the timed region), same frames and warp, packed output (6 W H)
                         "standalone_us": standalone_us,
                         "standalone_frac": (6 * rows * cols / (standalone_us * 1e-6)) / 1e9 / HBM_PEAK_GBS if standalone_us else None,
                         "standalone_valu_frac_spec": (valu_per_px_packed * rows * cols / (standalone_us * 1e-6)) / VALU_PEAK_SPEC if (standalone_us and valu_per_px_packed) else None},
            "roofline_secondary": secondary,
            "reference_kernel": reference_kernel,
            "configs": configs,
            "multi_stream": multi_stream,
            "multi_stream_field": multi_stream_field,
        }
        if world == 1 and K == 1 and not args.no_cpu_baseline:
            from tests import oracle_lib
            oracle = oracle_lib.load()
            # one thread per PHYSICAL core of the CPUs this process may run on (the NUMA node of the GPU when bound: 64 cores / 128 SMT threads on
            # the pool's 2 x EPYC 9575F boxes).  Measured in round 3: 128 threads give 7.5 frames/s, 64 give 10.5 -- the row-parallel stages
            # are memory-bound and the SMT siblings only add contention.
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            nthreads = max(1, min(ncpu // 2 if ncpu >= 16 else ncpu, 64))
            # timed with the -O3 -march=native build of the same sources (oracle/Makefile `fast`, built on this machine); the
            # parity oracle (-O2 -mavx2) stays the checker -- and checks this build's first frames.  No compiler here: the parity build is timed.
            fast = oracle_lib.load_fast()
            full, single, same = cpu_baseline(fast or oracle, rig, args.preset, nthreads, args.cpu_budget, args.format,
                                              lens_params if args.lens == "fused" else None, delay, check_against=oracle if fast else None)
            if same is False:
                raise SystemExit("bench.py: the -O3 timing build of the oracle does not reproduce the parity build's frames")
            result["cpu_baseline"] = {"value": full["value"], "unit": "frames/s", "cores": nthreads, "kind": "port",
                                      "per_core": full["value"] / nthreads, "p50_ms": full["p50_ms"], "p99_ms": full["p99_ms"], "stage_ms": full["stage_ms"],
                                      "single_thread": single,
                                      "build": ("oracle/Makefile `fast`: g++ -O3 -march=native -ffp-contract=off, built on this host; first frames equal to the -O2 -mavx2 parity build's: %s" % same)
                                               if fast else "oracle/Makefile default: g++ -O2 -mavx2 -mfma -ffp-contract=off (no timing build could be made here)",
                                      "sample": f"{full['frames']} consecutive steady-state frames of the same clip and settings ({full['cpu_seconds']:.1f} s of CPU work; oracle = CPU "
                                                f"restatement of the reference; 4:2:0 conversion, tracking-frame downscale, optical flow and remap "
                                                f"row / point-parallel over {nthreads} threads = one per physical core of the GPU's NUMA node), then "
                                                f"{single['frames'] if single else 0} frames on ONE thread (single_thread); stage_ms = mean ms per frame from the oracle's own timers"}
            if args.quality_frames > 0 and args.lens == "off" and clip is not None:
                try:
                    result["quality"] = quality_pass(lvk, ctx, oracle, clip, args.preset, max(args.quality_frames, 4 * delay + 2), nthreads)
                except Exception as e:
                    result["quality"] = {"error": repr(e)}
        print(json.dumps(result), flush=True)
        if n_ranks != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but {n_ranks} rank(s) reported\n")
            rc = 3
    for r in rigs:
        r.close()
    if world > 1:
        dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)
    return result


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--gpu-sampler":
        sampler_main(sys.argv[2:])
    else:
        main()
