# VERDICT r04 item 2a (GPU box, repo root): the adjoint's returning cursor atomic replaced by offsets[key] + a rank the forward pass's
# (now returning) counter atomic left per cached sample.  A/B of the bench step, product build against -DRF_EXP_TICKET, alternating.
# (the experiment's hooks live in a patch since round 6 -- the product source carries no development code: applied to a scratch copy here;
# the patch was cut against the round-5 source, re-cut it if it no longer applies)
mkdir -p /tmp/rf_exp && cp thr3ed_atom_amd/csrc/relu_field_kernels.hip /tmp/rf_exp/ && (cd /tmp/rf_exp && patch -p3 relu_field_kernels.hip < $OLDPWD/tools/experiments/r05_exp_ticket_gather_nop1.patch) || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include /tmp/rf_exp/relu_field_kernels.hip"
/opt/rocm/bin/hipcc $FLAGS -DRF_EXP_TICKET -o tools/exp_ticket.so || exit 1
for rep in 1 2 3; do
for lib in thr3ed_atom_amd/csrc/librelu_field_hip.so tools/exp_ticket.so; do
  RF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 --windows 0 --second-point-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
print('$lib', 'step %.4f ms' % l['ms_per_step'], 'psnr %.3f' % l['final_specular_psnr'], {k: round(v['avg_ms'], 4) for k, v in l['kernels'].items()})
"
done
done
