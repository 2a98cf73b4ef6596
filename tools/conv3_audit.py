"""Audit of the compiled k_conv3 kernels in their register-staged-loader form (experiment, see LAB_NOTES round 5): between the
loader's markers no compiler instruction may name the fixed queue registers v72..v167."""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
bad, inside, in_asm = [], False, False
reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
for ln, line in enumerate(text, 1):
    if "CV3_LOADER_BEGIN" in line: inside = True
    if "CV3_LOADER_END" in line: inside = False
    if ";;#ASMSTART" in line: in_asm = True
    if ";;#ASMEND" in line: in_asm = False; continue
    if not inside or in_asm: continue
    for m in reg.finditer(line.split(";")[0]):
        lo = int(m.group(1) or m.group(2)); hi = int(m.group(1) or m.group(3))
        if hi >= 72 and lo <= 167: bad.append((ln, line.strip()))
print(len(bad), "compiler instructions naming v72..v167 inside the loader regions")
sys.exit(1 if bad else 0)
