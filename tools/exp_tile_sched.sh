# A/B of the packet kernel's tile scheduling (run on the GPU box from the repo root): waves per workgroup x XCD-aware tile map
mkdir -p gpurun_out
OUT=gpurun_out/${1:-tile_sched}.txt
: > $OUT
for rep in 1 2; do
for wpb in 4 1; do for xr in 0 1; do
  RF_TILE_WPB=$wpb RF_TILE_XCD_ROWS=$xr python tools/frame_time.py 9 >> $OUT 2>/dev/null
done; done; done
cat $OUT
