mkdir -p gpurun_out/r03_cn2
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_contextnet_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r03_cn2/pytest.txt
timeout 200 python bench.py --model contextnet --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_cn2/bench_cn.json 2> gpurun_out/r03_cn2/bench_cn.err
TFASR_CN_WGRAD_GROUP=0 timeout 200 python bench.py --model contextnet --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_cn2/bench_cn_nogroup.json 2>> gpurun_out/r03_cn2/bench_cn.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_cn2/bench_M.json 2>> gpurun_out/r03_cn2/bench_cn.err
cat gpurun_out/r03_cn2/pytest.txt; cut -c1-330 gpurun_out/r03_cn2/*.json; tail -5 gpurun_out/r03_cn2/bench_cn.err
