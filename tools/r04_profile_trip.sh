#!/bin/bash
# Round-4 measurement trip: bench line, batch sweep, rocprofv3 kernel stats, PMC passes, side benches.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r04_bench_b8192.json; cut -c1-400 gpurun_out/r04_bench_b8192.json
echo "== sweep"; rm -f gpurun_out/r04_bench_sweep.jsonl
for b in 1024 3072 4096 8192 16384 32768 65536; do
  timeout 600 python bench.py --batch $b --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>&1 | tail -1 >> gpurun_out/r04_bench_sweep.jsonl
done
python - <<'PY'
import json
for line in open('gpurun_out/r04_bench_sweep.jsonl'):
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    print(d['config']['per_gpu_batch'], d['value'], d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})
PY
echo "== rocprof stats"
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $REPO/bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-profile --no-extras > $REPO/gpurun_out/rocprof_run.log 2>&1
cd $REPO
find /tmp/prof -name "*kernel_stats*" -exec cp {} gpurun_out/r04_rocprofv3_kernel_stats_b8192.csv \;
grep -E "k_|rocclr" gpurun_out/r04_rocprofv3_kernel_stats_b8192.csv | cut -c1-170
echo "== pmc"; bash tools/gpu_diag.sh 8192 2>&1 | tail -40 > gpurun_out/r04_pmc_mlp_b8192.txt; tail -40 gpurun_out/r04_pmc_mlp_b8192.txt
python tools/pmc_traffic_json.py 8192 && cp profiles/r04_pmc_hbm_traffic.json gpurun_out/
echo "== side benches"
timeout 600 python bench_loader.py 64 21600 2>&1 | tail -1 > gpurun_out/r04_loader_bench_highres.json
timeout 600 python bench_stream.py 2>&1 | tail -1 > gpurun_out/r04_stream_bench_highres.json
timeout 600 python bench_metrics.py 2>&1 | tail -1 > gpurun_out/r04_metrics_bench_scoring.json
timeout 600 python bench_online_mlp.py 2>&1 | tail -1 > gpurun_out/r04_online_mlp_bench.json
for f in loader_bench_highres stream_bench_highres metrics_bench_scoring online_mlp_bench; do echo $f; cut -c1-300 gpurun_out/r04_$f.json; done
echo "== trial groups"
timeout 300 python bench_hpo.py 2>&1 | tail -1 > gpurun_out/r04_hpo_bench.json; cut -c1-300 gpurun_out/r04_hpo_bench.json
echo "== CNN"; bash tools/r04_cnn_trip.sh 2>&1 | tail -30
echo "== chain stamps + A/B of this round's chain changes (same box)"
timeout 300 python tools/chain_stamps.py 8192 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r04_chain_stamps.txt; cat gpurun_out/r04_chain_stamps.txt
{ echo "k_chain_fb<32> at 8192 columns, same box, back to back, us per launch (bench.py --steps 100, per-kernel events scaled to the step):"
  echo "  trunk=0           : one weight queue per stage (round 3: CS_CHAIN_TRUNK=0)"
  echo "  trunk=1 ablate=128: the run of 512-wide stages as one continuous weight stream (chain_trunk), 128-wide stages repeated by both wave halves"
  echo "  trunk=1 ablate=0  : + the two wave halves split the contraction of the 128-wide stages, one epilogue (what ships)"
  for rep in 1 2; do for cfg in "0 0" "1 128" "1 0"; do set -- $cfg
  CS_CHAIN_TRUNK=$1 CS_CHAIN_ABLATE=$2 timeout 300 python bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trunk=$1 ablate=$2 step ms', d['ms_per_step'], {k:round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
  done; done; } > gpurun_out/r04_chain_ab.txt 2>&1; cat gpurun_out/r04_chain_ab.txt
echo "== de-phase probe"; timeout 120 tools/dephase_probe.bin 2>&1 | tee gpurun_out/r04_dephase_probe.txt
