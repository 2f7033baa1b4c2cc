"""hipcc pads no hazard whose producer is inside an inline-asm string; a VGPR written there and read as an MFMA A/B/C operand fewer than 2 wait
states later is read stale on gfx950 (this bit rollout16_kernel in round 3).  tools/check_asm_hazards.py scans the device assembly for the pair;
here: the scanner itself on hand-written listings, and the rollout kernels' real assembly (the other sources: run the tool, ~4 min)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
H = importlib.util.module_from_spec(spec)
spec.loader.exec_module(H)

BAD = """
_Z1kv:
	;;#ASMSTART
	v_cvt_pk_f16_f32 v41, v41, v68
	;;#ASMEND
	s_nop 0
	v_mfma_f32_16x16x16_f16 v[34:37], v[58:59], v[40:41], v[34:37]
"""
GOOD = BAD.replace("s_nop 0", "s_nop 1")
NATIVE = BAD.replace(";;#ASMSTART", "").replace(";;#ASMEND", "")     # the compiler's own instruction: its recognizer pads it, not our business
OTHER_REG = BAD.replace("v[40:41]", "v[42:43]")


def test_scanner_flags_the_pair_and_only_the_pair():
    assert len(H.scan_lines(BAD.splitlines())) == 1
    assert H.scan_lines(GOOD.splitlines()) == []
    assert H.scan_lines(NATIVE.splitlines()) == []
    assert H.scan_lines(OTHER_REG.splitlines()) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_rollout_kernels_have_no_asm_to_mfma_pair():
    assert H.scan(os.path.join(H.C, "kernels_rollout.hip"), []) == []
