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


# ---------------------------------------------------------------------------------------------------------------------------------------
# The wave-specialised update kernel's hand-over (kernels_update_mfma.hip: mg_post) publishes its LDS event counters WITHOUT an s_waitcnt:
# safe only while every access to the hand-over regions is a DS instruction (one wave's DS instructions execute in issue order; a FLAT
# access to LDS has no ordering against a later ds_write of the flag).  The regions reach mg_body as generic pointers, so DS-vs-FLAT depends
# on address-space inference after inlining -- nothing in the source guarantees it.  This test does: it disassembles the kernel out of
# the SHIPPED library and demands no flat_* / scratch_* instruction, no private segment and no spilled register.
# (-DMG_POST_WAIT builds the documented fallback with the wait.)
# ---------------------------------------------------------------------------------------------------------------------------------------
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "ppo-libtorch_amd", "libppo_hip.so")


def _device_code_objects(tmp):
    """llvm-objdump --offloading writes the bundles next to its input: work on a copy in a scratch directory."""
    lib = os.path.join(tmp, "lib.so")
    shutil.copy(LIB, lib)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))), reason="needs the built library and llvm-objdump")
def test_ws_update_kernel_handover_regions_are_ds_only():
    with tempfile.TemporaryDirectory() as tmp:
        seen = 0
        for co in _device_code_objects(tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            if "fwd_bwd_mfma_ws_kernel" not in notes:
                continue
            # kernel metadata: one YAML map per kernel (keys in alphabetical order, so .name comes before the counts we want)
            for block in notes.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                if "fwd_bwd_mfma_ws_kernel" not in name:
                    continue
                seen += 1
                assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)) == 0, name
                assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
                assert int(re.search(r"\.sgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            body, inside = [], False
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    inside = "fwd_bwd_mfma_ws_kernel" in m.group(1)
                    continue
                if inside:
                    body.append(line.strip())
            assert len(body) > 1000 and any(l.startswith("ds_") for l in body)
            bad = [l for l in body if re.match(r"^(flat_|scratch_)", l)]
            assert bad == [], bad[:5]
        assert seen >= 2, "both instantiations of fwd_bwd_mfma_ws_kernel (CartPole / masked MountainCar) must be in the shipped library"
