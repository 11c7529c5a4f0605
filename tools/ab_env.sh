# A/B of environment settings on the training step (GPU box, repo root): tools/ab_env.sh "VAR=a" "VAR=b" ...
for setting in "$@"; do
  for rep in 1 2; do
    env $setting python bench.py ${AB_STEPS:---steps 40 --warmup 10} --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
k=l['kernels']
print('$setting', 'step %.4f' % l['ms_per_step'], ' '.join('%s %.4f' % (n[-22:], v['avg_ms']) for n,v in k.items() if not n.startswith('(no')))
"
  done
done
