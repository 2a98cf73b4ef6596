#!/bin/bash
# The measurement trip behind profiles/rNN_*: bench line, batch sweep, rocprofv3 kernel stats, PMC passes, side benches.
# usage: gpurun -- bash tools/profile_trip.sh 05   (files land in gpurun_out/ as r05_*: copy what should be judged into profiles/)
R=r${1:-05}
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/${R}_bench_b8192.json; cut -c1-400 gpurun_out/${R}_bench_b8192.json
echo "== sweep"; rm -f gpurun_out/${R}_bench_sweep.jsonl
for b in 1024 3072 4096 8192 16384 32768 65536; do
  timeout 600 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>&1 | tail -1 >> gpurun_out/${R}_bench_sweep.jsonl
done
R=$R python - <<'PY'
import json, os
for line in open('gpurun_out/%s_bench_sweep.jsonl' % os.environ['R']):
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    print(d['config']['per_gpu_batch'], d['value'], d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})
PY
echo "== rocprof stats"
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $REPO/bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-profile --no-extras > $REPO/gpurun_out/rocprof_run.log 2>&1
cd $REPO
find /tmp/prof -name "*kernel_stats*" -exec cp {} gpurun_out/${R}_rocprofv3_kernel_stats_b8192.csv \;
grep -E "k_|rocclr" gpurun_out/${R}_rocprofv3_kernel_stats_b8192.csv | cut -c1-170
echo "== pmc"; bash tools/gpu_diag.sh 8192 2>&1 | tail -40 > gpurun_out/${R}_pmc_mlp_b8192.txt; tail -40 gpurun_out/${R}_pmc_mlp_b8192.txt
CS_ROUND=${1:-05} python tools/pmc_traffic_json.py 8192 && cp profiles/${R}_pmc_hbm_traffic.json gpurun_out/ 2>/dev/null
echo "== side benches"
timeout 600 python bench_loader.py 64 21600 2>&1 | tail -1 > gpurun_out/${R}_loader_bench_highres.json
timeout 600 python bench_stream.py 2>&1 | tail -1 > gpurun_out/${R}_stream_bench_highres.json
timeout 600 python bench_metrics.py 2>&1 | tail -1 > gpurun_out/${R}_metrics_bench_scoring.json
timeout 600 python bench_online_mlp.py 2>&1 | tail -1 > gpurun_out/${R}_online_mlp_bench.json
for f in loader_bench_highres stream_bench_highres metrics_bench_scoring online_mlp_bench; do echo $f; cut -c1-300 gpurun_out/${R}_$f.json; done
echo "== trial groups"
timeout 300 python bench_hpo.py 2>&1 | tail -1 > gpurun_out/${R}_hpo_bench.json; cut -c1-300 gpurun_out/${R}_hpo_bench.json
echo "== CNN"; bash tools/cnn_trip.sh ${1:-05} 2>&1 | tail -30
echo "== chain stamps"
timeout 300 python tools/chain_stamps.py 8192 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/${R}_chain_stamps.txt; cat gpurun_out/${R}_chain_stamps.txt
echo "== CNN weight-gradient stamps"
timeout 300 python tools/cnn_wgrad_stamps.py 512 2>&1 | grep -v amdgpu.ids | tail -16 > gpurun_out/${R}_cnn_wgrad_stamps_final.txt; head -8 gpurun_out/${R}_cnn_wgrad_stamps_final.txt
