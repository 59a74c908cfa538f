cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "nn 12096 1024 256 0 0 1" "nt 12096 256 1024 0 1 1" "tn 256 1024 12096 1 0 24" "big 65536 1000 640 0 0 1"; do
  set -- $cfg; tag=$1; shift
  timeout 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/lds_$tag -o x --output-format csv -- $R/tools/hwprobe/gemm_timing $@ > $R/gpurun_out/lds_$tag.log 2>&1
done
ls $R/gpurun_out/lds_nn | head
