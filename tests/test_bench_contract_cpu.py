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
