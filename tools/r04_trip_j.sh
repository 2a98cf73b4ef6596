#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== coop stamps 1024"; timeout 300 python tools/coop_stamps.py 1024 2>&1 | grep -v amdgpu.ids | tail -24 | tee gpurun_out/r04_j_coop_stamps_1024.txt
echo "== coop stamps 2048"; timeout 300 python tools/coop_stamps.py 2048 2>&1 | grep -v amdgpu.ids | tail -24 | tee gpurun_out/r04_j_coop_stamps_2048.txt
echo "== sysfs"; python - <<'PY' 2>&1 | tee gpurun_out/r04_j_sysfs.txt
import glob, os, torch
pr = torch.cuda.get_device_properties(0)
print([a for a in dir(pr) if 'pci' in a.lower() or 'uuid' in a.lower()])
for a in ('pci_domain_id','pci_bus_id','pci_device_id','uuid'):
    print(a, getattr(pr, a, None))
for c in sorted(glob.glob('/sys/class/drm/card*/device')):
    rp = os.path.realpath(c)
    fr = sorted(glob.glob(os.path.join(c, 'hwmon', 'hwmon*', 'freq1_input')))
    try: v = [open(f).read().strip() for f in fr]
    except Exception as e: v = [repr(e)]
    try: u = open(os.path.join(c, 'unique_id')).read().strip()
    except Exception as e: u = None
    print(c, rp, v, u)
PY
