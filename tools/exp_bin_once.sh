# "Bin once, use twice" go / no-go (GPU box, repo root): (a) today's forward pair, (b) the same launch without the specular feature gather
# (-DRF_EXP_NO_P1: the density-only march of both renders), (c) the brick-sorted feature gather (tools/exp_bin_once.py).
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include thr3ed_atom_amd/csrc/relu_field_kernels.hip"
/opt/rocm/bin/hipcc $FLAGS -DRF_EXP_NO_P1 -o tools/exp_nop1.so || exit 1
/opt/rocm/bin/hipcc $FLAGS -DRF_EXP_GATHER -o tools/exp_gather.so || exit 1
for lib in thr3ed_atom_amd/csrc/librelu_field_hip.so tools/exp_nop1.so; do
  RF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 --windows 0 --second-point-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
print('$lib', 'step %.4f ms' % l['ms_per_step'], {k: round(v['avg_ms'], 4) for k, v in l['kernels'].items()})
"
done
RF_LIB_PATH=$PWD/tools/exp_gather.so python tools/exp_bin_once.py 15
