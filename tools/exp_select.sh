# VERDICT r05 item 3 (GPU box, repo root): the batch drawn INSIDE the paired forward launch (no selection launch; the first workgroup clears
# the loss sums) -- tools/experiments/r06_select_in_kernel.patch applied to a scratch copy of the product source.  A/B of the bench step,
# product build against the patched build with RF_SELECT_IN_KERNEL=1, alternating, three pairs.
mkdir -p /tmp/rf_exp && cp thr3ed_atom_amd/csrc/relu_field_kernels.hip /tmp/rf_exp/ && (cd /tmp/rf_exp && patch -p3 relu_field_kernels.hip < $OLDPWD/tools/experiments/r06_select_in_kernel.patch) || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I include /tmp/rf_exp/relu_field_kernels.hip -o tools/exp_select.so || exit 1
for rep in 1 2 3; do
for lib in thr3ed_atom_amd/csrc/librelu_field_hip.so tools/exp_select.so; do
  RF_SELECT_IN_KERNEL=1 RF_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 --windows 2 --second-point-rays 0 --train256-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'ms_per_step %.4f' % d['ms_per_step'], 'windows', d['ms_per_step_windows']['min'], {k: round(v['avg_ms'],4) for k,v in d['kernels'].items()}, 'psnr', d['final_specular_psnr'])
"
done; done
