# forward kernels: waves per SIMD the register allocator is asked for (RF_FWD_WAVES) on the bench step (GPU box, repo root)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include thr3ed_atom_amd/csrc/relu_field_kernels.hip"
for w in 4 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DRF_FWD_WAVES=$w -o tools/exp_fwd.so || exit 1
  RF_LIB_PATH=$PWD/tools/exp_fwd.so python bench.py --steps 40 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 --windows 0 --second-point-rays 0 --timed-steps 10 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
print('fwd waves=$w', 'step %.4f' % l['ms_per_step'], 'fwd %.4f' % l['kernels']['render_forward[spec+diffuse,save]']['avg_ms'])
"
done
