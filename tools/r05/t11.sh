#!/bin/bash
# round 5, call 11: one launch per LSTM step (fused recurrent product + cell) on the prediction network's stream
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t11
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_lstm_persist_gpu.py tests/test_model_gpu.py tests/test_dp_gpu.py tests/test_rnnt_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_parity_baseline_gpu.py -m gpu -x -q -k "m_dims_greedy or T462_B4 or conformer_s_16" 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for v in 0 1 0 1; do
  TFASR_LSTM_FUSED_STEP=$v timeout 200 python bench.py $B > $O/single_f$v.json 2>> $O/err
  TFASR_LSTM_FUSED_STEP=$v timeout 200 python bench.py $B --dp-hooks > $O/dp_f$v.json 2>> $O/err
  echo "fused step=$v: single $(grep -o '"ms_per_step": [0-9.]*' $O/single_f$v.json) launches $(grep -o '"launches_per_step": [0-9.]*' $O/single_f$v.json) | dp $(grep -o '"ms_per_step": [0-9.]*' $O/dp_f$v.json)"
done
TFASR_LSTM_FUSED_STEP=1 timeout 200 python bench.py $B --model S > $O/S_f1.json 2>> $O/err
TFASR_LSTM_FUSED_STEP=0 timeout 200 python bench.py $B --model S > $O/S_f0.json 2>> $O/err
echo "S: fused $(grep -o '"ms_per_step": [0-9.]*' $O/S_f1.json) | pair $(grep -o '"ms_per_step": [0-9.]*' $O/S_f0.json)"
bash tools/prof_quick.sh r5_t11/prof_M > $O/prof.txt 2>&1
grep -E "lstm|total kernel" $O/prof.txt | cut -c1-160
tail -3 $O/err
