"""The bench line the driver parses: required keys and types, checked on the line committed under profiles/ (the output of
`python bench.py` on an MI355X) - a format regression in bench.py shows up here without a GPU."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    line = open(os.path.join(REPO, "profiles", "r01_bench_b8192.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    # value = columns of the timed steps / time
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]


def test_round3_bench_line_carries_the_repaired_fields():
    """The line committed at the end of round 3 (profiles/r03_bench_b8192.json): per-kernel times that add up to the step, the
    acceptance leg against the CPU restatement, the published model with its own CPU baseline, no dead ReLU head."""
    d = json.loads(open(os.path.join(REPO, "profiles", "r03_bench_b8192.json")).read().strip().splitlines()[-1])
    assert d["metric"] == "training columns/sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert sum(k["ms_per_step"] for k in d["kernels"].values()) <= d["ms_per_step"] * 1.001
    assert 0 < d["kernels_note"]["scaled_by"] <= 1.0
    assert d["coop_timeouts"] == 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0 and r["traffic_source"].startswith("profiles/r03_")
    a = d["heldout"]["against_cpu_restatement"]
    assert a["rel_diff_mae_all_outputs"] < 0.02 and a["min_R2"]["engine_bf16"] > 0.5 and a["min_R2"]["cpu_fp32"] > 0.5
    assert all(v is None or v > 0 for v in d["heldout"]["per_variable"]["R2"].values())        # no variable lost to a dead unit
    p = d["pub_mlp"]
    assert p["columns_per_s"] > 2e7 and p["cpu_baseline"]["kind"] == "port" and p["cpu_baseline"]["cores"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["unit"] == d["unit"]


def test_round4_bench_line_carries_the_sweep():
    """The line committed at the end of round 4 (profiles/r04_bench_b8192.json): SURVEY 8(d) config (2) names B in {1024, 8192, 65536} -
    the other two sizes ride inside the driver's line (`sweep`), the dead `gpu_busy_percent` field is gone, the traffic figure is this round's."""
    d = json.loads(open(os.path.join(REPO, "profiles", "r04_bench_b8192.json")).read().strip().splitlines()[-1])
    assert d["metric"] == "training columns/sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert d["config"]["per_gpu_batch"] == 8192 and "workload" in d["config"]
    sw = d["sweep"]
    assert set(sw) == {"1024", "65536"}
    for b, p in sw.items():
        assert p["per_gpu_batch"] == int(b) and p["blocks"] == 10 and p["unit"] == "columns/s"
        assert abs(p["value"] - int(b) / (p["ms_per_step"] * 1e-3)) <= 0.01 * p["value"]
        assert 0 < p["whole_step_frac"] < 1 and 0 < p["dominant_frac"] < 1 and p["dominant_kernel"].startswith("k_") and p["coop_timeouts"] == 0
    assert sw["65536"]["value"] > d["value"] > sw["1024"]["value"]
    assert "gpu_busy_percent" not in d["timing"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0 and r["traffic_source"].startswith("profiles/r04_")
    assert sum(k["ms_per_step"] for k in d["kernels"].values()) <= d["ms_per_step"] * 1.001
    assert d["stream"]["value"] > 0 and d["cnn"]["ms_per_step"] > 0 and d["cpu_baseline"]["kind"] == "port"
    # prediction in calls of 65536 rows (not bound by max_batch), the shader clock read from the card the process runs on
    assert d["predict"]["rows_per_call"] == 65536 and d["predict"]["columns_per_s"] > 2.5e8
    assert d["timing"]["gpu_sclk_mhz"]["card_matched_by"] == "pci address" and d["timing"]["gpu_sclk_mhz"]["median"] > 1000


def test_bench_defaults_and_flags():
    src = open(os.path.join(REPO, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    assert 'default=1)' in src.split('"--gpus"')[1][:60]            # no flags: one GPU


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the driver's command shape): the parent must start two ranks itself
    instead of exiting.  Without a GPU here every rank stops at the "needs a GPU" check - which proves the children ran
    with RANK / WORLD_SIZE set and that the parent returned their failure code."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "must be launched with torch.distributed.run" not in (r.stdout + r.stderr)
    assert (r.stdout + r.stderr).count("bench.py needs a GPU") >= 1


def test_failure_reporter_prints_one_error_line_when_rank_zero_is_blocked(tmp_path):
    """bench.py, N > 1 (round 4): a rank r > 0 that fails leaves its message and exits; the launcher then SIGTERMs rank 0, which may
    be blocked in a C call where Python-level handlers never run - its watcher thread on the signal wake-up pipe prints the ONE
    line (with "error" and the failed rank's message) and ends the process with a non-zero code.  No GPU involved."""
    import signal
    import subprocess
    import sys
    import time
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "import bench\n"
        "rank = int(sys.argv[1])\n"
        "rep = bench.FailureReporter(rank, 2)\n"
        "if rank == 1:\n"
        "    try:\n"
        "        raise RuntimeError('boom on rank 1')\n"
        "    except RuntimeError as e:\n"
        "        rep.fail(e)\n"
        "print('ready', flush=True)\n"
        "import ctypes\n"
        "libc = ctypes.CDLL(None)\n"
        "while True:\n"
        "    libc.sleep(60)\n"                     # (a signal ends one sleep(); the main thread goes straight back into C)
    )
    # both "ranks" share this test process as their parent (as ranks share the launcher), hence one report directory
    env = dict(os.environ, MASTER_PORT="45678")
    r1 = subprocess.run([sys.executable, str(child), "1"], capture_output=True, text=True, env=env, timeout=120)
    assert r1.returncode == 1 and "boom on rank 1" in r1.stderr
    p0 = subprocess.Popen([sys.executable, str(child), "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert p0.stdout.readline().strip() == "ready"
    time.sleep(0.2)
    p0.send_signal(signal.SIGTERM)
    out, _ = p0.communicate(timeout=60)
    assert p0.returncode == 1
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "SIGTERM" in d["error"] and "boom on rank 1" in d["failed_ranks"]["1"]


def test_acceptance_check_of_the_bench_line():
    """bench.acceptance_check (round 6: a statistic): per variable - and for the all-output MAE - the mean PAIRED difference of two
    sides over the data orders, held to 2 % + two standard errors; the first side's own scatter bounded by the second's.  Driven
    here on made-up tables: it passes where the difference is inside the allowance, fails where the first side is off by more,
    fails an erratic first side although its mean is right, and reports the fields the line promises."""
    import importlib
    import math
    import sys
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    cpu = [{"a": 1.00, "b": 2.00}, {"a": 1.04, "b": 2.02}, {"a": 0.98, "b": 1.99}, {"a": 1.01, "b": 2.01}]
    eng = [{"a": 1.03, "b": 2.01}, {"a": 0.99, "b": 2.03}, {"a": 1.02, "b": 2.00}, {"a": 1.00, "b": 2.02}]
    c = bench.acceptance_check({"engine_bf16": eng, "cpu_fp32": cpu})
    assert c["passed"] and c["margin"] > 0 and c["orders"] == 4 and c["sides"] == ["engine_bf16", "cpu_fp32"]
    assert set(c) >= {"engine_vs_cpu", "cpu_vs_cpu_other_order", "engine_vs_engine_other_order", "allowed", "worst_variable", "tolerance",
                      "order_to_order_sd", "per_variable_passed", "scatter_passed"}
    assert c["cpu_vs_cpu_other_order"]["a"] == round((1.04 - 0.98) / 0.98, 4)
    d = [0.03, -0.05, 0.04, -0.01]
    mean_c = (1.00 + 1.04 + 0.98 + 1.01) / 4
    md = sum(d) / 4
    se = math.sqrt(sum((x - md) ** 2 for x in d) / 3) / 2 / mean_c
    assert abs(c["engine_vs_cpu"]["of_the_order_means"]["a"] - abs(md) / mean_c) < 1e-3
    assert abs(c["engine_vs_cpu"]["signed"]["a"] - md / mean_c) < 1e-3
    assert abs(c["engine_vs_cpu"]["se"]["a"] - se) < 1e-3 and abs(c["allowed"]["a"] - (0.02 + 2 * se)) < 1e-3
    off = [{"a": 1.00, "b": 2.30}, {"a": 1.01, "b": 2.31}, {"a": 1.02, "b": 2.29}, {"a": 1.00, "b": 2.30}]   # b off by 15 %, scatter of ~1 %
    c = bench.acceptance_check({"engine_bf16": off, "cpu_fp32": cpu})
    assert not c["passed"] and not c["per_variable_passed"] and c["worst_variable"] == "b" and c["margin"] < 0
    # an erratic first side: the mean of b is right, its scatter is ten times the second side's - the standard error widens the bar
    # (per variable passes), the scatter bound does not
    wild = [{"a": 1.00, "b": 1.60}, {"a": 1.04, "b": 2.45}, {"a": 0.98, "b": 1.75}, {"a": 1.01, "b": 2.22}]
    c = bench.acceptance_check({"engine_bf16": wild, "cpu_fp32": cpu})
    assert c["per_variable_passed"] and not c["scatter_passed"] and not c["passed"]
    # the all-output MAE is held to the same rule: 2 % + two standard errors of its own paired differences
    c = bench.acceptance_check({"engine_bf16": eng, "cpu_fp32": cpu}, {"engine_bf16": [1.0, 1.01, 1.0, 1.0], "cpu_fp32": [1.0, 1.0, 1.0, 1.0]})
    assert c["passed"] and c["all_outputs"]["rel_diff_of_the_order_means"] == 0.0025 and c["all_outputs"]["signed"] == 0.0025
    assert abs(c["all_outputs"]["allowed"] - (0.02 + 2 * 0.0025)) < 1e-4           # sd of (0, .01, 0, 0) = .005, / sqrt(4)
    c = bench.acceptance_check({"engine_bf16": eng, "cpu_fp32": cpu}, {"engine_bf16": [1.05, 1.06, 1.05, 1.04], "cpu_fp32": [1.0, 1.0, 1.0, 1.0]})
    assert not c["passed"] and not c["all_outputs_passed"] and c["per_variable_passed"]
    # any two sides of the leg can be compared (the line compares engine / bf16 emulation / float32 pairwise)
    c = bench.acceptance_check({"x": eng, "y": cpu, "z": off}, None, ("x", "y"))
    assert c["sides"] == ["x", "y"] and c["passed"]
