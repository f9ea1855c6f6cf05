#!/bin/bash
# Same-box A/B of the working tree against a git revision:  tools/experiments/ab_worktree.sh build [rev]   (here; default HEAD)
# builds build_variants/v1.so from `rev` (csrc only) and v2.so from the working tree;  tools/experiments/abl.sh run  then times
# both on the frozen scene in ONE gpurun call (boxes differ by a few per cent, launches on one box by ~0.5 %).
set -u
REV=${2:-HEAD}
rm -rf build_variants; mkdir -p build_variants /tmp/ab_keep
cp -r starst3r_amd/csrc /tmp/ab_keep/csrc_work
git stash -q -- starst3r_amd/csrc include 2>/dev/null; STASHED=$?
git checkout -q $REV -- starst3r_amd/csrc include
python -m starst3r_amd.build --force > /dev/null 2>&1 || echo "build of $REV failed"
cp starst3r_amd/libst3r_hip.so build_variants/v1.so; echo "$REV" > build_variants/v1.txt
git checkout -q HEAD -- starst3r_amd/csrc include
[ $STASHED -eq 0 ] && git stash pop -q
python -m starst3r_amd.build --force > /dev/null 2>&1 || echo "build of the working tree failed"
cp starst3r_amd/libst3r_hip.so build_variants/v2.so; echo "working tree" > build_variants/v2.txt
git status --short starst3r_amd/csrc include
