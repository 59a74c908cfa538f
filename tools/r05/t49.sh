#!/bin/bash
# round 5, call 49: same-box A/B of the query backward's score read-back (production: four aligned 4-byte LDS reads; probe lib: previous commit)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t49
mkdir -p $O
cd $R
for i in 1 2; do
bash tools/prof_quick.sh r5_t49/new$i > $O/new$i.txt 2>&1
TFASR_LIB=$R/tools/hwprobe/libtfasr_probe.so bash tools/prof_quick.sh r5_t49/old$i > $O/old$i.txt 2>&1
done
for f in new1 old1 new2 old2; do echo "$f: $(grep 'bwd_qT' $O/$f.txt | sed 's/(.*`//' | cut -c1-70) | bwd_k $(grep 'bwd_k_kernel' $O/$f.txt | sed 's/.*` | 160 | [0-9.]* | //' | cut -c1-6)"; done
