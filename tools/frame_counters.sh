# counter passes over the frame render only (run on the GPU box from the repo root): tools/frame_counters.sh <tag> [cfg1|cfg4]
# -> gpurun_out/cnt_<tag>/summary.md: one column per launch of render_frame_tile_kernel (the first = the warm-up frame)
set -x
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-frame}
CFG=${2:-cfg1}
OUT=$ROOT/gpurun_out/cnt_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/tools/frame_only.py $CFG 3 > $OUT/p$i.out 2> $OUT/p$i.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/tools/frame_only.py $CFG 5 > $OUT/stats.out 2> $OUT/stats.err
cd $ROOT
python tools/summarize_counters.py $OUT "render_frame_tile_kernel" > $OUT/summary.md
python tools/summarize_rocprof.py $OUT/stats >> $OUT/summary.md 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.md
