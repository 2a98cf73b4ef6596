#!/bin/bash
# scratch: streamed training with the next chunk's loader in slices inside the steps' gaps
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stream_gpu.py -x -q 2>&1 | tail -5
O=gpurun_out/r05_stream_gaps.txt; : > $O
for rep in 1 2; do
for cfg in "main opt 0" "gaps opt 0" "gaps opt 4" "gaps chain 0" "gaps chain 4" "side opt 0"; do
  set -- $cfg
  echo "== CS_STREAM_LOADER=$1 CS_STREAM_GAP=$2 CS_STREAM_SLICES=$3" >> $O
  CS_STREAM_LOADER=$1 CS_STREAM_GAP=$2 CS_STREAM_SLICES=$3 timeout 300 python bench_stream.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ratio_to_train_only','ratio_to_serial_sum','loader_share_hidden')}, d.get('passes',{}).get('stream_ms'))" >> $O
done; done
cat $O
