# A/B of library variants on the training step (GPU box, repo root): tools/ab_bench.sh tools/exp_a.so tools/exp_b.so ...
for lib in "$@"; do
  for rep in 1 2; do
    RF_LIB_PATH=$PWD/$lib python bench.py --steps 40 --warmup 10 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
k=l['kernels']
print('$lib', 'step %.4f' % l['ms_per_step'], ' '.join('%s %.4f' % (n.split('[')[0][-12:]+n[n.find('['):] if '[' in n else n[-14:], v['avg_ms']) for n,v in k.items() if not n.startswith('(no')))
"
  done
done
