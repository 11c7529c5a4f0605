# rocprofv3 passes of the bench (run on the GPU box from the repo root): kernel trace + stats, then the PMC passes separately
set -x
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 1"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/tcc.err
cd $ROOT
python tools/summarize_rocprof.py $OUT/trace > $OUT/kernel_stats.md
python tools/summarize_pmc.py $OUT/fetch $OUT/write $OUT/tcc > $OUT/pmc.md
python tools/make_pmc_traffic.py $OUT/fetch $OUT/write --gt-frames 8 --frames-per-leg 3 > $OUT/pmc_traffic.json
find $OUT -name "*.csv" -size +3M -delete
cat $OUT/kernel_stats.md | head -30; cat $OUT/pmc.md; du -sh $OUT
