O=gpurun_out/glds3.txt
: > $O
timeout 60 ./tools/hwprobe/gemm_big_test 343552 640 1000 2>&1 | grep -E "joint|differ|worst" >> $O
for t in 512 300; do
echo "=== bench M BIG_T=$t" >> $O
TFASR_GEMM_BIG_T=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-260 >> $O
done
echo "=== bench S" >> $O
timeout 200 python bench.py --model S --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-260 >> $O
timeout 800 python -m pytest tests -m gpu -q 2>&1 | tail -3 >> $O
cat $O
