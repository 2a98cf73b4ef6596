#!/bin/bash
# Diagnostics: clocks under load + PMC counters for the engine kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== idle clocks"; rocm-smi --showclocks --showperflevel --showpower 2>&1 | grep -E "sclk|mclk|fclk|Perf|Power|socclk" | head
python bench.py --batch 65536 --steps 3000 --warmup 10 --cpu-budget 0 --no-profile > gpurun_out/diag_bench.log 2>&1 &
BP=$!
sleep 12
echo "== clocks under load"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | head
wait $BP; tail -c 400 gpurun_out/diag_bench.log; echo
cd /tmp
echo "== pmc pass 1"
rm -rf /tmp/pmc1; timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc1 -- python3 $REPO/bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-profile > $REPO/gpurun_out/pmc1.log 2>&1
find /tmp/pmc1 -name "*counter_collection*" -exec cp {} $REPO/gpurun_out/pmc1_counters.csv \;
cd $REPO
python - <<'PY'
import csv, collections
try:
    rows=list(csv.DictReader(open('gpurun_out/pmc1_counters.csv')))
except Exception as e:
    print('no pmc csv', e); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'][:40]
    if not (k.startswith('void k_') or k.startswith('k_')): continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k, {c:round(sum(v)/len(v),1) for c,v in d.items()}, 'n=',len(next(iter(d.values()))))
PY
tail -3 gpurun_out/pmc1.log | cut -c1-300
