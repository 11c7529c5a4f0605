# A/B of the persistent brick pass against one workgroup per brick, in the driver's window (GPU box, repo root)
for rep in 1 2 3; do
for v in 1 0; do
    RF_BRICK_PERSIST=$v python bench.py --steps 20 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
k=l['kernels']
print('persist=$v', 'step %.4f' % l['ms_per_step'], ' '.join('%s %.4f' % (n[-22:], v['avg_ms']) for n,v in k.items() if not n.startswith('(no')))
"
done
done
