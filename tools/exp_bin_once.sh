# "Bin once, use twice" go / no-go (GPU box, repo root): (a) today's forward pair, (b) the same launch without the specular feature gather
# (-DRF_EXP_NO_P1: the density-only march of both renders), (c) the brick-sorted feature gather (tools/exp_bin_once.py).
# (the experiment's hooks live in a patch since round 6 -- the product source carries no development code: applied to a scratch copy here;
# the patch was cut against the round-5 source, re-cut it if it no longer applies)
mkdir -p /tmp/rf_exp && cp thr3ed_atom_amd/csrc/relu_field_kernels.hip /tmp/rf_exp/ && (cd /tmp/rf_exp && patch -p3 relu_field_kernels.hip < $OLDPWD/tools/experiments/r05_exp_ticket_gather_nop1.patch) || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include /tmp/rf_exp/relu_field_kernels.hip"
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
