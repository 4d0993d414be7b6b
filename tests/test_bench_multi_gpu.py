"""The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one rank per GPU, gloo barrier, no collective on the data
path -- SURVEY.md section 8e "replicas of independent streams"), exercised functionally on a 1-GPU box: LVK_BENCH_SHARE_GPU=1 maps the
ranks onto the GPUs present.  No scaling claim comes out of this (two ranks share one GPU); it checks the contract line: ONE JSON line,
n_gpus 2, two distinct streams (clip seeds), whole-job value = the frames of all ranks / the slowest rank's time."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_a_shared_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, LVK_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--rows", "1080", "--cols", "1920", "--pool", "64"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["warmup"] == 5 and r["scaling"] == "weak" and r["unit"] == "frames/s"
    ranks = r["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1]
    assert ranks[0]["clip_seed"] != ranks[1]["clip_seed"]                       # two DIFFERENT streams, not one stream twice
    assert all(x["frames"] == 20 for x in ranks)
    slowest = max(x["elapsed_s"] for x in ranks)
    assert abs(r["value"] - 40 / slowest) <= 0.02 * r["value"]                  # aggregate = sum of the frames / max over ranks of the time
    assert abs(r["ms_per_step"] - slowest / 20 * 1e3) <= 0.02 * r["ms_per_step"]
    assert r["sustained"]["frames"] >= 1200
    print("\n[bench --gpus 2 on a shared GPU] value %.0f frames/s; per rank: %s" % (r["value"], [(x["device"], x["numa_cpus"], round(x["frames_per_s"])) for x in ranks]))
