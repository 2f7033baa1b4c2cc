#!/usr/bin/env python3
"""What ties a committed profile summary to the binary a bench run loads: a fingerprint of the SOURCES libppo_hip.so is built from (every kernel file, the
internal headers, the C-ABI header, the Makefile with its flags).  The built library is not in git and its bytes could differ with the build path; its
sources are what both a collection run and a later bench run see.  tools/collect_profiles.sh writes profiles/<tag>_meta.json with this fingerprint;
bench.py emits a profile-derived field only when the newest set's fingerprint equals the running tree's.

    python tools/src_fingerprint.py                 -> the fingerprint
    python tools/src_fingerprint.py --meta TAG      -> writes profiles/TAG_meta.json"""
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    c = os.path.join(ROOT, "ppo-libtorch_amd", "csrc")
    return sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.hpp")) + [os.path.join(c, "Makefile"), os.path.join(ROOT, "include", "ppo_hip.h")])


def source_fingerprint():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def file_sha16(path):
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--meta":
        tag = sys.argv[2]
        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=20).stdout.strip() or None
        except Exception:
            head = None
        if not head:
            # the GPU box's snapshot carries no .git: tools/stamp_head.sh (run in the container before gpurun) leaves `git rev-parse HEAD` (+ "-dirty"
            # when the tree had uncommitted changes) in .build_head, which travels with the snapshot
            try:
                with open(os.path.join(ROOT, ".build_head")) as fh:
                    head = fh.read().strip() or None
            except OSError:
                head = None
        meta = {"tag": tag, "src_sha16": source_fingerprint(), "lib_sha16": file_sha16(os.path.join(ROOT, "ppo-libtorch_amd", "libppo_hip.so")), "git_head": head,
                "collected_unix": int(time.time()), "files": sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", tag + "_*")))}
        with open(os.path.join(ROOT, "profiles", tag + "_meta.json"), "w") as fh:
            json.dump(meta, fh, indent=1)
        print(json.dumps(meta))
    else:
        print(source_fingerprint())


if __name__ == "__main__":
    main()
