#!/bin/bash
# Leaves the commit the tree stands on in .build_head (git-ignored; travels to the GPU box, whose snapshot has no .git) so that
# tools/src_fingerprint.py --meta can record it:  tools/stamp_head.sh && gpurun -- tools/collect_profiles.sh <tag>
cd "$(dirname "$0")/.."
H=$(git rev-parse HEAD)
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then H="$H-dirty"; fi
echo "$H" > .build_head
echo "$H"
