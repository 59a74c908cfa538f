#!/bin/bash
# round 5, call 14: joint projection with transposed accumulators (8-byte row pieces stored from registers, no LDS strip)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_t14
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_rnnt_gpu.py tests/test_gemm_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_parity_baseline_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "T462_B4 or conformer_s_16 or step" 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for v in 0 1 0 1; do
  TFASR_BIG_TR=$v timeout 200 python bench.py $B > $O/s$v.json 2>> $O/err
  python - <<PY
import json
d=json.loads(open("$O/s$v.json").read().strip().splitlines()[-1])
print("TR=$v: ms/step", d["ms_per_step"], "| joint fwd", d["roofline"]["ms_per_launch"], "ms, frac", d["roofline"]["frac"], "| rnnt", d["roofline_rnnt"]["ms_per_launch"])
PY
done
tail -2 $O/err
