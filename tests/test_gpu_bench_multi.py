"""The N > 1 paths of bench.py, rehearsed on ONE GPU: two fresh child processes per case (`torch.distributed.run`, backend
gloo because RCCL refuses two ranks on one device, ITA_FORCE_DEVICE=0), a few steps each, and the JSON line rank 0 prints
is checked for what the driver reads (`n_gpus`, `scaling`, the frame counts, `gather_every`).  This is the code the
driver's 2 / 4 / 8-GPU run executes with backend nccl; no hardware scaling curve comes out of it.

Three processes touch the GPU at a time (pytest + two ranks), well inside the box's limit of six."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, steps=8):
    env = dict(os.environ, ITA_FORCE_DEVICE="0", ITA_BENCH_SETTLE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", str(steps),
           "--warmup", "8", "--backend", "gloo", "--no-latency"] + extra
    r = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]          # rank 0 prints ONE line
    return json.loads(lines[0])


def test_weak_scaling_line_per_step_gather():
    """default: 1024 frames per GPU is too slow a rehearsal; 300 frames per rank, one stream, one all-gather per step"""
    out = _run(["--frames-per-gpu", "300"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 8
    cfg = out["config"]
    assert cfg["frames_per_gpu"] == 300 and cfg["global_batch"] == 600 and cfg["parallelism"] == "dp2"
    assert cfg["gather_every"] == 1 and "every 1 step" in cfg["gather"]
    assert out["value"] > 0 and abs(out["value"] - 600 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"] is None


def test_weak_scaling_line_batched_gather():
    out = _run(["--frames-per-gpu", "300", "--gather-every", "8"])
    assert out["config"]["gather_every"] == 8 and out["scaling"] == "weak" and out["config"]["global_batch"] == 600


def test_strong_scaling_uneven_shards_graph_schedule():
    """config 4 cut over the GPUs with a total that does not divide: 257 frames -> shards of 129 and 128, which at
    <= 256 frames per GPU run the HIP-graph schedule (one velocity buffer per replay, staged into the gather's own send
    buffers; rank 1's rows padded to rank 0's)"""
    out = _run(["--global-batch", "257"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["global_batch"] == 257 and cfg["frames_per_gpu"] == 129 and cfg["gather_every"] == 8
    assert "HIP-graph" in cfg["schedule"]
    assert abs(out["value"] - 257 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]


def test_strong_scaling_uneven_shards_one_stream():
    out = _run(["--global-batch", "601", "--schedule", "stream"])
    assert out["scaling"] == "strong" and out["config"]["frames_per_gpu"] == 301 and out["config"]["gather_every"] == 1
