"""The continuous weight stream of the wide chain (csrc/chainw.h: cws_tile) keeps its queue in v[192:255] and the sign-mask fetch in
v[190:191] - registers the compiler does not know are in use.  This test disassembles the gfx950 code object of the library AS BUILT
and fails if any instruction other than the stream's own asm statements touches v[190:255] in a k_chainw* kernel (tools/chainw_audit.py).
No GPU needed."""
import importlib.util
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_queue_registers_are_left_alone_by_compiled_code(capsys):
    from climsim_amd import build as b
    lib = b.build()
    spec = importlib.util.spec_from_file_location("chainw_audit", os.path.join(REPO, "tools", "chainw_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    breaks = mod.audit(mod.disassemble(lib))
    out = capsys.readouterr().out
    assert breaks == 0, out
    lines = [ln for ln in out.splitlines() if ln.startswith("_Z")]
    assert len(lines) >= 5, out                                   # k_chainw<0|1>, k_chainw_fb, the two grouped forms
    for ln in lines:
        m = re.search(r"highest VGPR outside the queue statements v(-?\d+), queue statements (\d+), breaks 0", ln)
        assert m, ln
        assert int(m.group(2)) >= 20 and int(m.group(1)) < 190, ln
