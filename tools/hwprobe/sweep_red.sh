cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sw_*
for cfg in "256 64" "256 96" "256 128" "256 192" "256 256" "256 384" "512 64" "512 128" "512 192"; do
  set -- $cfg
  TFASR_RED_THREADS=$1 TFASR_RED_GRID=$2 timeout 100 rocprofv3 --kernel-trace --stats -d gpurun_out/sw_$1_$2 -- python tools/probe_kernels.py pointwise > /dev/null 2>&1
done
