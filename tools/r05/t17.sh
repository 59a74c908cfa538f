#!/bin/bash
# round 5, call 17: the round's kept bench lines, kernel tables (as run / in line), PMC traffic and MFMA utilisation
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_final
mkdir -p $O
cd $R
timeout 500 python bench.py > $O/bench_M.json 2> $O/bench_M.err
timeout 300 python bench.py --model S > $O/bench_S.json 2> $O/bench_S.err
timeout 200 python bench.py --dp-hooks --no-cpu-baseline --no-extras > $O/bench_M_dp_hooks.json 2>> $O/bench_M.err
TFASR_DP_FORCE_SPLIT=1 timeout 200 python bench.py --dp-hooks --no-cpu-baseline --no-extras > $O/bench_M_dp_hooks_split.json 2>> $O/bench_M.err
timeout 200 python bench.py --model S --dp-hooks --no-cpu-baseline --no-extras > $O/bench_S_dp_hooks.json 2>> $O/bench_S.err
timeout 200 python bench.py --model S-streaming --no-cpu-baseline --steps 50 > $O/bench_S_streaming.json 2>> $O/bench_S.err
timeout 200 python bench.py --mode decode --model S --steps 10 --warmup 2 > $O/decode_S.json 2> $O/decode.err
timeout 200 python bench.py --mode decode --model M --steps 10 --warmup 2 > $O/decode_M.json 2>> $O/decode.err
timeout 200 python bench.py --mode ctc-decode --steps 5 --warmup 2 > $O/ctc_decode.json 2>> $O/decode.err
timeout 300 python bench.py --model contextnet --alpha 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/contextnet_L.json 2> $O/contextnet.err
bash tools/prof_quick.sh r5_final/prof_M > /dev/null 2>&1
( export TFASR_WGRAD_STREAM=0 TFASR_NO_PRED_STREAM=1 TFASR_DPEXT_AUX=0 TFASR_DEFER_SIDE=0; bash tools/prof_quick.sh r5_final/prof_M_inline > /dev/null 2>&1 )
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*.db' | head -1)" $O/pmc_traffic.json 30 > $O/pmc_top.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cd $R
bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/mfma_busy.txt $O/mfma_busy.txt
for f in bench_M bench_M_dp_hooks bench_M_dp_hooks_split bench_S bench_S_dp_hooks bench_S_streaming decode_S decode_M ctc_decode contextnet_L; do echo "== $f: $(grep -o '"ms_per_step": [0-9.]*' $O/$f.json | head -1) $(grep -o '"value": [0-9.e-]*' $O/$f.json | head -1)"; done
head -14 $O/prof_M_inline/stats.md | cut -c1-140
