"""Condense bench.py JSON lines (stdin) to one short line each; extra argv words are echoed as a tag."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        print(line[:300])
        continue
    k = {n: round(v["ms_per_step"], 4) for n, v in (d.get("kernels") or {}).items()}
    rf = d.get("roofline") or {}
    print(" ".join(sys.argv[1:]), "B=%d" % d["config"]["per_gpu_batch"], "%.2f Mcol/s" % (d["value"] / 1e6),
          "%.4f ms" % d["ms_per_step"], k, "dom=%s frac=%s" % (rf.get("kernel"), rf.get("frac")),
          "cpu=%s" % ((d.get("cpu_baseline") or {}).get("value")))
