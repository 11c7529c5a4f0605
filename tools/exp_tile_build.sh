# packet kernel variants by compile flag (GPU box, repo root): waves per SIMD
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include thr3ed_atom_amd/csrc/relu_field_kernels.hip"
for w in 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DRF_TILE_WAVES=$w -o tools/exp_tile_w$w.so || exit 1
  for rep in 1 2; do
  RF_FRAME_TILES=1 RF_LIB_PATH=$PWD/tools/exp_tile_w$w.so python bench.py --steps 3 --warmup 2 --cpu-rays 0 --dropin-steps 0 --windows 0 --second-point-rays 0 --highres-frames 3 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
f=l['fwd_render']; h=l['highres_render']
print('waves=$w', {k:(round(f[k]['ms_per_frame'],3), round(f[k]['kernel_ms_per_frame'],3)) for k in ('init_field','traversal')}, {k:round(v,3) for k,v in h.items() if isinstance(v,float) and k.startswith('ms_')})
"
  done
done
