# frame render only: timing + the issue counters of the frame kernel (run on the GPU box from the repo root): tools/frame_quick.sh <tag>
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-fq}
OUT=$ROOT/gpurun_out/fq_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --cpu-rays 0 --dropin-steps 0 --render-frames 5 --images 1"
python $ROOT/bench.py $ARGS > $OUT/line.json 2> $OUT/line.err
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python $ROOT/bench.py $ARGS --highres-frames 0 --render-frames 1 > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/p2 -- python $ROOT/bench.py $ARGS --highres-frames 0 --render-frames 1 > /dev/null 2> $OUT/p2.err
cd $ROOT
python tools/summarize_counters.py $OUT > $OUT/summary.md
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.md
python - <<PY
import json
d=json.load(open("$OUT/line.json"))
f=d["fwd_render"]
for k in ("init_field","traversal"):
    print(k, {x: round(f[k][x],4) for x in ("ms_per_frame","kernel_ms_per_frame")})
print("highres", d.get("highres_render",{}).get("ms_per_frame_occupancy_mask"), d.get("highres_render",{}).get("ms_per_frame_no_mask"))
print("errors", d.get("errors"))
PY
