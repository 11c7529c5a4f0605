# A/B of library variants in the DRIVER's window (steps 5-25 from the initialisation, 5 timed steps): tools/ab_driver.sh lib.so ...
for rep in 1 2 3; do
for lib in "$@"; do
    RF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
k=l['kernels']
print('$lib', 'step %.4f' % l['ms_per_step'], ' '.join('%s %.4f' % (n[-22:], v['avg_ms']) for n,v in k.items() if not n.startswith('(no')))
"
done
done
