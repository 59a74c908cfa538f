#!/bin/bash
# round 5, call 8: conv1/BN backward with the row's gradient prefetched, new parity cases, decode calibration, in-line kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv1_gram_gpu.py tests/test_model_gpu.py tests/test_reference_wiring_gpu.py -m gpu -x -q -s 2>&1 | grep -E "wiring\]|passed|failed|Error|error" | tail -30
timeout 1500 python -m pytest tests/test_parity_baseline_gpu.py -m gpu -x -q -s -k "T743_B1_L16 or m_dims_greedy" 2>&1 | grep -E "\[g1\]|passed|failed|Error|error|assert" | tail -20
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for g in 0 1024 768 512; do
  if [ $g = 0 ]; then unset TFASR_CONV1_GRID; else export TFASR_CONV1_GRID=$g; fi
  timeout 200 python bench.py $B > $O/single_g$g.json 2>> $O/err
  echo "conv1 grid=$g: $(grep -o '"ms_per_step": [0-9.]*' $O/single_g$g.json)"
done
unset TFASR_CONV1_GRID
timeout 300 python bench.py --mode decode --model M --steps 10 --warmup 2 > $O/decode_M.json 2>> $O/err
cut -c1-1500 $O/decode_M.json
export TFASR_WGRAD_STREAM=0 TFASR_NO_PRED_STREAM=1 TFASR_DPEXT_AUX=0
bash tools/prof_quick.sh r5_t8/prof_inline > $O/prof_inline.txt 2>&1
head -44 $O/prof_inline.txt | cut -c1-150
tail -3 $O/err
