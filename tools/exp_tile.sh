# Frame render A/B (GPU box, repo root): the ray-packet kernel (RF_FRAME_TILES=1 forces it) against the per-ray kernel (=0), configs[1] and configs[4].
run() {
  RF_FRAME_TILES=$1 python bench.py --steps 3 --warmup 2 --cpu-rays 0 --dropin-steps 0 --windows 0 --second-point-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
f=l['fwd_render']; h=l['highres_render']
print('tiles=$1', {k:(round(f[k]['ms_per_frame'],3), round(f[k]['kernel_ms_per_frame'],3)) for k in ('init_field','traversal')}, h and {k:round(v,3) for k,v in h.items() if isinstance(v,float) and k.startswith('ms_')}, h and h['mask_vs_no_mask_bit_identical'])
"
}
run 1; run 0; run 1; run 0
