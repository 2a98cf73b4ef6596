"""bench.py end to end on the GPU box: the one-GPU line, and the N = 2 flow (parent starts its own ranks, barriers, MAX over
ranks, weak value + strong leg + comm figures, ONE JSON line from rank 0) with both ranks sharing cuda:0 over gloo
(CS_BENCH_SHARE_GPU=1: the pool has one GPU per box and RCCL cannot put two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--steps", "5", "--warmup", "2", "--min-seconds", "0.05", "--rows", "65536", "--cpu-budget", "0", "--no-extras"]


def run(args, env_extra=None):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_one_gpu_line():
    d = run(FAST)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["strong"] is None and d["comm"] is None
    assert d["value"] == pytest.approx(d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert d["timing"]["blocks"] >= 1 and d["timing"]["steps_per_block"] == 5
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic_source"].startswith("profiles/")
    assert d["heldout"]["mse"] > 0 and d["predict"]["columns_per_s"] > 0


def test_two_ranks_start_themselves_and_report_weak_strong_and_comm():
    d = run(["--gpus", "2", *FAST, "--no-profile", "--batch", "2048"], {"CS_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4096
    assert d["value"] == pytest.approx(4096 / (d["ms_per_step"] * 1e-3), rel=1e-3)          # whole-job columns / max-over-ranks time
    s = d["strong"]
    assert s["scaling"] == "strong" and s["global_batch"] == 8192 and s["per_gpu_batch"] == 4096 and s["value"] > 0
    c = d["comm"]
    assert c["nranks"] == 2 and c["bytes"] == 4787200 and c["allreduce_us_per_step"] > 0
