# issue / L1 counters of the training-step kernels (run on the GPU box from the repo root): tools/step_counters.sh <tag> [extra bench args]
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-sc}
shift
OUT=$ROOT/gpurun_out/sc_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 10 --warmup 3 --cpu-rays 0 --dropin-steps 0 --render-frames 0 --highres-frames 0 --images 2 $*"
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p$i.err
done
cd $ROOT
python tools/summarize_pmc.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.md
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.md
