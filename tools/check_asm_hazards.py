#!/usr/bin/env python3
"""hipcc pads no hazard whose producer sits inside an inline-asm string (cdna_hip_programming.md 5.7 item 2): a VGPR written by an asm VALU
instruction and read as an MFMA A/B/C operand fewer than 2 wait states later is read STALE (seen on gfx950: rollout logits off by 1e-3).
This scans the device assembly of every kernel source for that pair and exits 1 if it finds one.
    tools/check_asm_hazards.py [file.hip ...]      (default: every csrc/*.hip)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "ppo-libtorch_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden".split() + ["-I" + os.path.join(ROOT, "include"), "-I" + C]
NOSLP = {"kernels_update_mfma.hip", "kernels_gemm.hip", "kernels_generic_fused.hip"}
NEED = 2


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path, extra):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + (["-fno-slp-vectorize"] if os.path.basename(path) in NOSLP else []) + extra + ["--cuda-device-only", "-S", "-o", f.name, path]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(f.name).read().splitlines()
    return scan_lines(lines, path)


def scan_lines(lines, path="<asm>"):
    found, in_asm, func = [], False, "?"
    pending = []   # [written regs, states elapsed, text, line]
    for n, raw in enumerate(lines, 1):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if re.match(r"^[A-Za-z_][\w.$]*:", s) and not s.startswith(".L"):
            func, pending = s.rstrip(":"), []
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            for t in toks[1:4]:
                for w, st, text, ln in pending:
                    if st < NEED and regs(t) & w:
                        found.append("%s:%d %s  <- %d state(s) after asm `%s` (line %d) in %s" % (os.path.basename(path), n, s, st, text, ln, func))
        states = int(toks[0]) + 1 if op == "s_nop" else 1
        pending = [[w, st + states, text, ln] for w, st, text, ln in pending if st + states < NEED]
        if in_asm and op.startswith("v_") and toks:
            pending.append([regs(toks[0]), 0, s, n])
    return found


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(C, "*.hip")))
    bad = []
    for p in files:
        bad += scan(p, [])
    for b in bad:
        print(b)
    print("%d asm-VALU -> MFMA operand pair(s) closer than %d wait states in %d file(s)" % (len(bad), NEED, len(files)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
