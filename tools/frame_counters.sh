# counter passes over the frame render only (run on the GPU box from the repo root): tools/frame_counters.sh <tag>
set -x
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-frame}
OUT=$ROOT/gpurun_out/cnt_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 2 --images 1"
cd /tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/p$i.err
done
cd $ROOT
python tools/summarize_counters.py $OUT > $OUT/summary.md
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.md
