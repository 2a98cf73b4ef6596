"""Audit of the wide chain's kernels AS SHIPPED (CPU only: the gfx950 code object inside climsim_amd/libclimsim_hip.so, disassembled
with llvm-objdump): the weight queue of the continuous stream lives in v[192:255] and the sign-mask fetch in v[190:191] - registers
only the asm statements of chainw.h name.  Prints, per kernel, the highest VGPR any OTHER instruction touches and the lines that
break the reservation (none expected).  `python tools/chainw_audit.py [listing.s]`; exit status 1 on a break."""
import os
import re
import struct
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESERVED_FROM = 190
Q = r"(19[2-9]|2[0-5]\d)"
ALLOWED = (
    re.compile(r"^\s*global_load_dwordx4 v\[" + Q + r":" + Q + r"\], v\[\d+:\d+\], off$"),
    re.compile(r"^\s*global_load_dwordx2 v\[190:191\], v\[\d+:\d+\], off$"),
    re.compile(r"^\s*v_mfma_f32_32x32x16_bf16 v\[\d+:\d+\], v\[" + Q + r":" + Q + r"\], v\[\d+:\d+\], v\[\d+:\d+\]$"),
    re.compile(r"^\s*v_mov_b32(_e32)? v\d+, v19[01]$"),
)
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def kernels(asm_text):
    """(name, instruction lines) of every k_chainw* function of an llvm-objdump listing or of a hipcc -S listing"""
    cur, body = None, []
    for line in asm_text.splitlines():
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w*k_chainw\w*)>?:", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur and (re.match(r"^[0-9a-f]+ <", line) or line.strip().startswith(".amdhsa_kernel") or line.startswith(".Lfunc_end")):
            yield cur, body
            cur = None
            continue
        if cur:
            body.append(line.split("//")[0])
    if cur:
        yield cur, body


def audit(asm_text):
    bad_total = 0
    for name, body in kernels(asm_text):
        top, bad, n_queue = -1, [], 0
        for raw in body:
            line = raw.split(";")[0].rstrip()
            if not line.strip() or line.strip().startswith((".", "s_")) or line.endswith(":"):
                continue
            regs = []
            for m in VREG.finditer(line):
                if m.group(1):
                    regs.append(int(m.group(1)))
                else:
                    regs.extend((int(m.group(2)), int(m.group(3))))
            if not regs:
                continue
            if any(a.match(line) for a in ALLOWED):
                n_queue += 1
                other = [r for r in regs if r < RESERVED_FROM]
                top = max([top] + other)
                continue
            hi = max(regs)
            top = max(top, hi)
            if hi >= RESERVED_FROM:
                bad.append(line.strip())
        print(f"{name}: highest VGPR outside the queue statements v{top}, queue statements {n_queue}, breaks {len(bad)}")
        for b in bad[:10]:
            print("   ", b)
        bad_total += len(bad)
    return bad_total


def disassemble(so_path):
    """the gfx950 code object of a HIP shared library (clang offload bundle in .hip_fatbin) -> llvm-objdump listing"""
    d = open(so_path, "rb").read()
    i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in " + so_path
    n = struct.unpack_from("<Q", d, i + 24)[0]
    off, blob = i + 32, None
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", d, off)
        off += 24
        triple = d[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            blob = d[i + o:i + o + sz]
    assert blob, "no gfx950 code object in " + so_path
    tmp = so_path + ".audit.co"
    try:
        open(tmp, "wb").write(blob)
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
        return subprocess.run([objdump, "-d", "--no-show-raw-insn", tmp], capture_output=True, text=True, check=True).stdout
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def main():
    lib = os.environ.get("CLIMSIM_HIP_LIB") or os.path.join(REPO, "climsim_amd", "libclimsim_hip.so")
    return 1 if audit(disassemble(lib)) else 0


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(1 if audit(open(sys.argv[1]).read()) else 0)
    sys.exit(main())
