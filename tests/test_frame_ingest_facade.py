"""lvk::FrameIngest of the C++ facade (include/lvk/FrameIngest.hpp = the plugin's Interop/FrameIngest, Modules/OBS-Plugin/Interop/FrameIngest.cpp:36-142)
with a stand-in obs_source_frame: Select / upload_obs_frame / download_ocl_frame for every format against the oracle's bytes.  CPU: it compiles
against the header alone; GPU: it runs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "frame_ingest_facade.cpp")
FORMATS = {"I420": 1, "NV12": 2, "YVYU": 3, "YUY2": 4, "UYVY": 5, "RGBA": 6, "BGRA": 7, "BGRX": 8, "I444": 10, "BGR3": 11, "I422": 12, "I40A": 13,
           "I42A": 14, "YUVA": 15, "AYUV": 16}


def _build(tmp_path):
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    exe = str(tmp_path / "frame_ingest_facade")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", exe, SRC,
                           "-L" + os.path.join(ROOT, "livevisionkit_amd"), "-llvk_hip", "-L" + tlib, "-l:libamdhip64.so",
                           "-Wl,-rpath," + os.path.join(ROOT, "livevisionkit_amd"), "-Wl,-rpath," + tlib])
    return exe


def test_facade_frame_ingest_compiles(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_facade_frame_ingest_every_format(tmp_path, oracle):
    exe = _build(tmp_path)
    for name, fmt in FORMATS.items():
        for (rows, cols), pad in (((36, 52), 0), ((270, 480), 12)):
            rng = np.random.default_rng(fmt * 1000 + rows)
            planes = [rng.integers(0, 256, sh, dtype=np.uint8) for sh in oracle.obs_plane_shapes(name, rows, cols)]
            src = tmp_path / "planes.bin"
            with open(src, "wb") as f:
                for p in planes:
                    f.write(p.tobytes())
            prefix = str(tmp_path / f"out_{name}")
            r = subprocess.run([exe, str(fmt), str(rows), str(cols), str(pad), str(src), prefix], capture_output=True, text=True)
            assert r.returncode == 0 and "ok" in r.stdout, (name, rows, cols, r.stdout, r.stderr)
            want = oracle.ingest_obs(name, planes)
            got = np.fromfile(prefix + ".frame", np.uint8).reshape(rows, cols, 3)
            assert np.array_equal(got, want), (name, rows, cols)
            back = oracle.egress_obs(name, want, planes=[np.full_like(p, 0x5A) for p in planes])
            raw = np.fromfile(prefix + ".planes", np.uint8)
            off = 0
            for b in back:
                assert np.array_equal(raw[off:off + b.size].reshape(b.shape), b), (name, rows, cols)
                off += b.size


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["UYVY", "I444", "I420", "BGR3"])
def test_facade_ingest_filter_download_chain(tmp_path, oracle, name):
    """upload_obs_frame -> StabilizationFilter::apply(std::move(frame), frame) -> download_ocl_frame, the plugin's asynchronous path with the facade's
    classes (Interop/VisionFilter.cpp:151-253 around VSFilter.cpp:352-364), against the oracle's ingest -> filter -> egress."""
    from tests import oracle_lib, synth
    exe = _build(tmp_path)
    rows, cols, n, delay = 270, 480, 12, 3
    clip, _ = synth.make_clip(rows, cols, n, seed=53, jitter=1.0)
    fmt = FORMATS[name]
    ffmt = {"BGR3": 0}.get(name, 4)
    s = oracle_lib.preset("homography", predictive_samples=delay, min_scene_quality=0.3, min_tracking_quality=0.2)
    ost = oracle_lib.OracleStabilizer(oracle, s)
    want = []
    with open(tmp_path / "clip.bin", "wb") as f:
        for i, fr in enumerate(clip):
            planes = oracle.egress_obs(name, fr)
            for p in planes:
                f.write(p.tobytes())
            w, wts = ost.push(oracle.ingest_obs(name, planes), ts=i, fmt=ffmt)
            if w is not None:
                want.append(np.concatenate([p.reshape(-1) for p in oracle.egress_obs(name, w, planes=[np.full_like(p, 0x5A) for p in planes])]))
    r = subprocess.run([exe, "--stream", str(fmt), str(rows), str(cols), str(n), str(delay), str(tmp_path / "clip.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and f"stream ok: {len(want)} frames" in r.stdout, (r.stdout, r.stderr)
    got = np.fromfile(tmp_path / "out.bin", np.uint8)
    assert got.size == sum(w.size for w in want)
    off = 0
    for k, w in enumerate(want):
        assert np.array_equal(got[off:off + w.size], w), (name, k)
        off += w.size
