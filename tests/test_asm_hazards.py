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


# ---------------------------------------------------------------------------------------------------------------------------------------
# bwd_layer_kernel (kernels_generic_bwd.hip) keeps two tiles of LDS-DMA in flight across its barriers with COUNTED waits (s_waitcnt vmcnt(N)).  The DMAs are
# inline asm: the compiler's own s_waitcnt bookkeeping does not see them.  The design holds only while the tile loop has NO vector-memory load into a register:
# one the compiler emits itself (a scratch reload after a spill, a flat access) gets a wait that ignores the asm operations around it and drains the ring or
# worse, and one issued by asm leaves a register the compiler may copy BEFORE the counted wait that covers it -- which is what this test found in round 6 in
# bwd_layer_kernel<1, false, 2> (a v_mov of the index register in front of the wait; harmless on that path by luck), whereupon the index list's entries were
# moved to an LDS-DMA of their own.  Nothing in the source enforces any of this; tools/bwd_isa.sh looks by hand.  This test looks at EVERY instantiation in the
# SHIPPED library and fails the build when a toolchain change breaks one of them.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _functions(dis):
    out, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if name is not None and line.strip():
            out[name].append(line.split("//")[0].strip())
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))), reason="needs the built library and llvm-objdump")
def test_bwd_layer_kernel_tile_loops_hold_no_compiler_visible_vector_load():
    with tempfile.TemporaryDirectory() as tmp:
        seen = set()
        for co in _device_code_objects(tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            if "bwd_layer_kernel" not in notes:
                continue
            for block in notes.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                if "bwd_layer_kernel" not in name:
                    continue
                assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)) == 0, name
                assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
                assert int(re.search(r"\.sgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            for name, body in _functions(dis).items():
                if "bwd_layer_kernel" not in name:
                    continue
                m = re.search(r"bwd_layer_kernelILi(\d+)ELb(\d)ELi(\d+)E", name)
                p1 = m.group(2) == "1"
                seen.add(m.groups())
                assert [l for l in body if re.match(r"^(scratch_|flat_|buffer_)", l)] == [], name
                dma = [i for i, l in enumerate(body) if l.startswith("global_load_lds_dwordx4")]
                barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
                assert dma and barriers, name
                first_dma, last_barrier = dma[0], barriers[-1]
                # from the first DMA to the last barrier NO vector-memory load into a register at all: tiles and (layer 0) the index list's entries travel by
                # LDS-DMA (global_load_lds_dwordx4 / _dword) and come back through ds_reads, which the compiler does see
                loads = [(i, l) for i, l in enumerate(body[first_dma:last_barrier], first_dma) if re.match(r"^global_load_(?!lds_dword)", l)]
                assert loads == [], (name, loads)
                idx_dma = [l for l in body[first_dma:last_barrier] if l.startswith("global_load_lds_dword ")]
                assert (len(idx_dma) >= 1) == (not p1), (name, idx_dma)
                # the loop's waits on the vector-memory counter are the counted ones of the source (vmcnt(N) lgkmcnt(0), N = DMA pieces per tile [+ 1 store]) and
                # the drains of the last iterations -- a compiler-made wait would carry no lgkmcnt(0) partner or another count
                waits = [l for l in body[first_dma:last_barrier] if l.startswith("s_waitcnt") and "vmcnt" in l]
                counted = sorted({int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in waits})
                assert counted[0] == 0 and 1 <= len(counted) <= 3, (name, waits)
        assert len(seen) == 9, seen   # NB in {1, 4, 8} x {P1, layer 0 with one / two column blocks}
