#!/bin/bash
# round 5, call 9: FFModule forward tile height by tile count (32-row tiles when one round of 2 per CU holds them), M-dims greedy test
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t9
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_conv1_gram_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_parity_baseline_gpu.py -m gpu -x -q -s -k "m_dims_greedy" 2>&1 | grep -E "\[g1\]|passed|failed|Error|error|assert" | tail -8
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for v in 2 0 4 2 0; do
  if [ $v = 0 ]; then unset TFASR_FFN_VARIANT; else export TFASR_FFN_VARIANT=$v; fi
  timeout 200 python bench.py $B > $O/single_v$v.json 2>> $O/err
  echo "ffn variant=$v: $(grep -o '"ms_per_step": [0-9.]*' $O/single_v$v.json)"
done
unset TFASR_FFN_VARIANT
export TFASR_WGRAD_STREAM=0 TFASR_NO_PRED_STREAM=1 TFASR_DPEXT_AUX=0
bash tools/prof_quick.sh r5_t9/prof_inline > $O/prof_inline.txt 2>&1
grep -E "ffn_fused|conv1_bn|total kernel" $O/prof_inline.txt | cut -c1-150
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_t8/decode_M.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['tokens_emitted'], d['breakdown'])
PY
tail -3 $O/err
