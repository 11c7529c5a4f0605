# emit pair kernel: chunks in flight (RF_EMIT_G) x waves per SIMD (RF_EMIT_WAVES) on the bench step (GPU box, repo root)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include thr3ed_atom_amd/csrc/relu_field_kernels.hip"
for cfg in "2 6" "1 8" "1 6" "2 5" "3 5" "2 7" "2 6"; do
  set -- $cfg
  /opt/rocm/bin/hipcc $FLAGS -DRF_EMIT_G=$1 -DRF_EMIT_WAVES=$2 -o tools/exp_emit.so || exit 1
  RF_LIB_PATH=$PWD/tools/exp_emit.so python bench.py --steps 40 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 --windows 0 --second-point-rays 0 --timed-steps 10 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
print('G=$1 waves=$2', 'step %.4f' % l['ms_per_step'], 'emit %.4f' % l['kernels']['render_backward_emit_direct[spec+diffuse]']['avg_ms'], 'fwd %.4f' % l['kernels']['render_forward[spec+diffuse,save]']['avg_ms'], 'brick %.4f' % l['kernels']['brick_accumulate']['avg_ms'])
"
done
