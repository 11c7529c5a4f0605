# quick rocprofv3 passes of the training step only (run on the GPU box from the repo root): usage tools/profile_quick.sh <tag> [pmc counters...]
set -x
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-quick}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 0 ${BENCH_EXTRA}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err
cd $ROOT
python tools/summarize_rocprof.py $OUT/trace > $OUT/kernel_stats.md
python tools/summarize_pmc.py $OUT/fetch $OUT/write > $OUT/pmc.md
find $OUT -name "*.csv" -size +3M -delete
head -20 $OUT/kernel_stats.md; cat $OUT/pmc.md
