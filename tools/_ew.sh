R=$(pwd)
O=$R/gpurun_out/ew5.txt
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_contextnet_gpu.py -m gpu -q -x 2>&1 | tail -2 > $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ewp -- python $R/tools/ew_bench.py > /tmp/ewp.log 2>&1
DB=$(find /tmp/ewp -name "*.db" | head -1)
python $R/tools/prof_summary.py "$DB" /tmp/ewp.md > /dev/null 2>&1
python $R/tools/_fmt.py /tmp/ewp.md "dwconv" >> $O
cd $R
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' >> $O
timeout 200 python bench.py --model contextnet --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' >> $O
cat $O
