"""bench.py's roofline block quotes counter figures (HBM traffic, VALU instructions per pixel, the kernel's clock, where the waves' time goes) from
the rocprofv3 summaries committed under profiles/ -- separate --pmc runs of the same command, never measured by the timed run.  Every key it
reads there must exist in the committed files (a missing key is reported as null by bench.py, never replaced by a constant: round-5 VERDICT found
a round-1 default of 531 instructions per pixel standing in for a key no file had), and the files must describe the kernels the pipeline runs."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_every_profile_key_bench_reads_is_committed():
    b = _bench()
    for fname in b.PROFILE_FILES["traffic"]:
        t = json.load(open(os.path.join(ROOT, "profiles", fname)))
        for k in b.PROFILE_KEYS["traffic"]:
            assert t.get(k) is not None, f"profiles/{fname}: {k} missing"
        assert t["rows"] == 2160 and t["cols"] == 3840 and t["kernel"].startswith("k_remap_") and "_420" in t["kernel"]
        assert 300 < t["valu_per_px"] < 700 and 300 < t["valu_per_px_packed"] < 700
        # traffic ~ algorithmic bytes (4.5 W H): no wasted re-reads
        assert 0.8 < t["hbm_bytes_per_launch"] / (4.5 * 3840 * 2160) < 1.2
    s = json.load(open(os.path.join(ROOT, "profiles", b.PROFILE_FILES["stalls"])))
    for kernel in ("k_remap_homography_420", "k_remap_mesh_420"):
        rec = s["live"][kernel]
        for k in b.PROFILE_KEYS["stalls"]:
            assert rec.get(k) is not None, f"profiles/{b.PROFILE_FILES['stalls']}: live.{kernel}.{k} missing"
        assert 1500 < rec["clock_mhz_under_kernel"] < 2600
        assert abs(rec["active_inst_any_frac"] + rec["wait_inst_any_frac"] + rec["wait_any_frac"] - 1.0) < 0.02
        assert 0.3 < rec["valu_issue_slot_frac"] <= 1.0
    assert "k_remap_homography_420" in s["alone"]


def test_bench_has_no_constant_standing_in_for_a_counter():
    """No numeric fall-back where a counter figure is read, and no peak whose denominator is a probe's clock."""
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"\.get\(\"valu_per_px[a-z_]*\",\s*[0-9]", text)
    assert "VALU_PEAK_MEASURED" not in text and "valu_frac_measured" not in text
    for k in ("clock_mhz_under_kernel", "valu_busy_frac", "valu_frac_at_kernel_clock"):
        assert k in text


def test_aggregate_of_the_multi_gpu_legs():
    """bench.py --gpus N > 1: whole-job figures of the legs every rank runs at once = the frames of all ranks / the SLOWEST rank's time; per-rank
    figures kept; a leg that failed on one rank is reported as an error, a leg that did not run as null (pure function: no GPU needed)."""
    b = _bench()

    def host(v, s, p50, p99, node):
        return {"value": v, "frames": 600, "elapsed_s": s, "GBps_each_way": v * 12.44 / 1e3, "latency_ms": {"p50": p50, "p99": p99}, "pinned_planes_numa_node": node}

    def cfg(v, s, p50, p99):
        return {"value": v, "steps": 400, "frames": 400, "elapsed_s": s, "p50_ms": p50, "p99_ms": p99}
    reports = [{"rank": 0, "numa_node": 0, "pcie": {"current_link_width": "16"}, "host_fed": host(3000.0, 0.20, 0.52, 0.55, 0), "config4_1080p": cfg(11000.0, 0.036, 0.11, 0.14)},
               {"rank": 1, "numa_node": 1, "pcie": {"current_link_width": "16"}, "host_fed": host(2400.0, 0.25, 0.60, 0.71, 1), "config4_1080p": {"workload": "x", "error": "boom"}}]
    a = b.aggregate_legs(reports)
    h = a["host_fed"]
    assert h["frames"] == 1200 and h["slowest_rank_s"] == 0.25 and abs(h["value"] - 1200 / 0.25) < 1e-9
    assert h["per_rank_frames_per_s"] == [3000.0, 2400.0] and h["p99_ms"] == 0.71 and h["p50_ms"] == 0.60 and h["per_rank_planes_numa_node"] == [0, 1]
    assert len(h["per_rank_GBps_each_way"]) == 2
    assert "error" in a["config4_1080p"] and a["config5_4k_lens"] is None and a["ranks"] is reports
