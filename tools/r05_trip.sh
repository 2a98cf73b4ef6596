#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cnn_gpu.py -x -q 2>&1 | tail -3
O=gpurun_out/r05_cnn_wgrad_ab.txt; : > $O
for rep in 1 2 3; do
for cfg in "0 4 0 S4" "1 4 0.8 -" "1 5 0.6 -" "0 4 0.5 -" "1 4 0.6 -" "1 5 0.8 -"; do
  set -- $cfg
  unset CS_CNN_WGRAD_SPLITS
  export CS_CW2_PERSIST=$1 CS_CNN_WGRAD_ROUNDS=$2 CS_CNN_WGRAD_TAPER=$3
  [ $4 = S4 ] && export CS_CNN_WGRAD_SPLITS=4
  echo -n "persist $1 rounds $2 taper $3 $4: " >> $O
  timeout 300 python bench_cnn.py 512 60 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $O
done; done
sort $O | awk '{k=$1" "$2" "$3" "$4" "$5" "$6" "$7; s[k]=s[k]" "$NF} END{for(k in s) print k, s[k]}' | sort
