# rocprofv3 passes of the bench (run on the GPU box from the repo root): tools/profile_bench.sh <tag>
#   1. the default bench line, un-profiled (what the driver runs)          -> gpurun_out/prof_<tag>/bench_default_line.json
#   2. kernel trace + stats; 3./4. the PMC passes (FETCH_SIZE, WRITE_SIZE) separately, as the microarchitecture guide prescribes;
#   5. TCC hit/miss                                                         -> kernel_stats.md, pmc.md, pmc_traffic.json
set -x
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_default_line.json 2> $OUT/bench_default.err
ARGS="--steps 20 --warmup 5 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 1 --windows 0 --second-point-rays 0 --train256-steps 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/tcc.err
cd $ROOT
python tools/summarize_rocprof.py $OUT/trace > $OUT/kernel_stats.md
python tools/summarize_pmc.py $OUT/fetch $OUT/write $OUT/tcc > $OUT/pmc.md
python tools/make_pmc_traffic.py $OUT/fetch $OUT/write --gt-frames 8 --frames-per-leg 2 --bench-args "$ARGS" > $OUT/pmc_traffic.json
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/kernel_stats.md | head -30; cat $OUT/pmc.md; du -sh $OUT; python tools/benchsum.py $OUT/bench_default_line.json
