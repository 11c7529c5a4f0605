# WRITE_SIZE / FETCH_SIZE against known byte counts in record-shaped store patterns (GPU box, repo root) -> gpurun_out/<tag>.md
export TMPDIR=/tmp
ROOT=$PWD
TAG=${1:-write_size_calibration}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in WRITE_SIZE FETCH_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -- $ROOT/tools/write_size_calibration > $OUT/$n.out 2> $OUT/$n.err
done
cd $ROOT
python tools/write_size_table.py $OUT > $OUT.md
cat $OUT.md
