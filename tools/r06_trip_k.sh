#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python bench.py --steps 5 --warmup 2 --min-seconds 0.05 --rows 65536 --cpu-budget 3 --no-profile --extras pub_mlp > gpurun_out/r06_bench_small.json 2> gpurun_out/r06_bench_small.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_small.json').read().strip().splitlines()[-1])
a=d["heldout"]["against_cpu_restatement"]
if "error" in a: print(a)
else:
    print("seconds", a["seconds"], "rel_diff_mae_all_outputs", a["rel_diff_mae_all_outputs"], "min_R2", a["min_R2"])
    for k in ("check","bf16_vs_fp32","engine_vs_fp32"):
        c=a[k]; print(k, "passed" , c.get("passed"), c["per_variable_passed"], c["all_outputs_passed"], "margin", c["margin"], c["worst_variable"])
        print("   signed", c["engine_vs_cpu"]["signed"]); print("   se", c["engine_vs_cpu"]["se"]); print("   all", c["all_outputs"])
PY
tail -3 gpurun_out/r06_bench_small.err
