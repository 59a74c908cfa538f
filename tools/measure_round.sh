#!/bin/bash
# Every bench line + profile of a round -> gpurun_out/$1/ (usage through gpurun: bash tools/measure_round.sh r03_final)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_M.json 2> $O/bench_M.err
timeout 300 python bench.py --model S > $O/bench_S.json 2> $O/bench_S.err
timeout 200 python bench.py --model S-streaming --no-cpu-baseline --steps 50 > $O/bench_S_streaming.json 2>> $O/bench_S.err
timeout 200 python bench.py --dp-hooks --no-cpu-baseline --no-extras > $O/bench_M_dp_hooks.json 2>> $O/bench_M.err
timeout 200 python bench.py --model S --dp-hooks --no-cpu-baseline --no-extras > $O/bench_S_dp_hooks.json 2>> $O/bench_S.err
timeout 200 python bench.py --mode decode --model S --steps 10 --warmup 2 > $O/decode_S.json 2> $O/decode.err
timeout 200 python bench.py --mode decode --model M --steps 10 --warmup 2 > $O/decode_M.json 2>> $O/decode.err
timeout 200 python bench.py --mode ctc-decode --steps 5 --warmup 2 > $O/ctc_decode.json 2>> $O/decode.err
timeout 300 python bench.py --model contextnet --alpha 2 --steps 20 --warmup 3 --no-cpu-baseline > $O/contextnet_L.json 2> $O/contextnet.err
python tools/lstm_bench.py 32 111 640 > $O/lstm_persist.txt 2>&1
python tools/lstm_bench.py 32 65 320 >> $O/lstm_persist.txt 2>&1
python tools/nccl_selfcheck.py > $O/nccl_selfcheck.txt 2>&1
bash tools/prof_quick.sh $TAG/prof_M > /dev/null 2>&1
bash tools/prof_quick.sh $TAG/prof_S --model S > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py "$(find $O/pmc_FETCH_SIZE -name '*.db' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*.db' | head -1)" $O/pmc_traffic.json 30 > $O/pmc_top.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for f in bench_M bench_M_dp_hooks bench_S bench_S_dp_hooks bench_S_streaming decode_S decode_M ctc_decode contextnet_L; do echo "== $f"; cut -c1-420 $O/$f.json; done
cat $O/lstm_persist.txt; tail -2 $O/nccl_selfcheck.txt
