# A/B of the brick shapes of the fused optimizer pass in the driver's window (GPU box, repo root): 4 x 8 x 8 (default) against 8^3
for rep in 1 2 3; do
for v in default 8; do
    if [ $v = default ]; then unset RF_BRICK_SIZE; else export RF_BRICK_SIZE=$v; fi
    python bench.py --steps 20 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
k=l['kernels']
print('brick=$v', 'step %.4f' % l['ms_per_step'], ' '.join('%s %.4f' % (n[-22:], v['avg_ms']) for n,v in k.items() if not n.startswith('(no')))
"
done
done
