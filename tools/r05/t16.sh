#!/bin/bash
# round 5, call 16: transposed accumulators in every 256-column big-GEMM variant (joint projection, conv2 products)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t16
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_rnnt_gpu.py tests/test_gemm_gpu.py tests/test_conv1_gram_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_parity_baseline_gpu.py -m gpu -x -q -k "T462_B4 or conformer_s_16" 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for v in 0 1 0 1; do
  TFASR_BIG_TR=$v timeout 200 python bench.py $B > $O/s$v.json 2>> $O/err
  TFASR_BIG_TR=$v timeout 200 python bench.py $B --model S > $O/S$v.json 2>> $O/err
  python - <<PY
import json
d=json.loads(open("$O/s$v.json").read().strip().splitlines()[-1]); e=json.loads(open("$O/S$v.json").read().strip().splitlines()[-1])
print("TR=$v: M ms/step", d["ms_per_step"], "| joint fwd", d["roofline"]["ms_per_launch"], "ms, frac", d["roofline"]["frac"], "| S ms/step", e["ms_per_step"])
PY
done
export TFASR_WGRAD_STREAM=0 TFASR_NO_PRED_STREAM=1 TFASR_DPEXT_AUX=0 TFASR_DEFER_SIDE=0
bash tools/prof_quick.sh r5_t16/prof_inline > $O/prof_inline.txt 2>&1
grep -E "gemm_big|total kernel" $O/prof_inline.txt | cut -c1-150
tail -2 $O/err
