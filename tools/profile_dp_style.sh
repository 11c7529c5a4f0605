# PMC passes of the data-parallel-style step on one GPU (bench.py --dp-style-step): adds brick_accumulate[sh2], render_backward[diffuse]
# and adam_step to profiles/pmc_traffic.json (run on the GPU box from the repo root, after tools/profile_bench.sh)
set -x
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r02_dp
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --cpu-rays 0 --dropin-steps 0 --highres-frames 0 --render-frames 0 --dp-style-step"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err
cd $ROOT
python tools/summarize_rocprof.py $OUT/trace > $OUT/kernel_stats.md
python tools/summarize_pmc.py $OUT/fetch $OUT/write > $OUT/pmc.md
python tools/make_pmc_traffic.py $OUT/fetch $OUT/write --merge-into profiles/pmc_traffic.json --source "profiles/r02c_dp_style_pmc.md (bench.py --dp-style-step)" > $OUT/pmc_traffic.json
find $OUT -name "*.csv" -size +3M -delete
head -16 $OUT/kernel_stats.md; cat $OUT/pmc.md; python tools/benchsum.py $OUT/trace.json
