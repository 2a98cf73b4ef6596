"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_diag.sh into profiles/rNN_pmc_hbm_traffic.json (NN = $CS_ROUND, default 05; what
bench.py's `roofline.traffic` reads).  Units and correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
counters in KiB, FETCH_SIZE counts half the bytes of wide coalesced streams on gfx950 -> traffic = (2*FETCH + WRITE) KiB.
    python tools/pmc_traffic_json.py 8192 [65536 ...]"""
import collections
import csv
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "profiles", f"r{os.environ.get('CS_ROUND', '05')}_pmc_hbm_traffic.json")


def norm(name):
    m = re.match(r"(?:void )?k_chain<\d+, (true|false), (?:true|false)>", name)
    if m:
        return f"k_chain<BM,{m.group(1)}>"
    if re.match(r"(?:void )?k_chain_fb<", name):
        return "k_chain_fb<BM>"
    if name.startswith("k_wgrad") or name.startswith("void k_wgrad"):
        return "k_wgrad"
    if name.startswith("k_optimizer"):
        return "k_optimizer"
    return None


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = norm(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
for b in sys.argv[1:]:
    f = per_kernel(os.path.join(REPO, "gpurun_out", f"pmc_fetch_b{b}.csv"), "FETCH_SIZE")
    w = per_kernel(os.path.join(REPO, "gpurun_out", f"pmc_write_b{b}.csv"), "WRITE_SIZE")
    doc[str(b)] = {k: {"FETCH_SIZE_KiB": round(f[k], 1), "WRITE_SIZE_KiB": round(w.get(k, 0.0), 1),
                       "traffic_bytes": int((2 * f[k] + w.get(k, 0.0)) * 1024)} for k in f}
    print(b, doc[str(b)])
json.dump(doc, open(OUT, "w"), indent=1)
