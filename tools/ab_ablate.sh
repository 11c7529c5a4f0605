# Ablations of the brick pass WITHOUT instrumentation (GPU box, repo root): builds a -DRF_BRICK_ABLATE copy of the library and times the
# pass inside bench.py's training step with parts switched off ($RF_BRICK_STAGGER: 0x100000 no tile loop, 0x200000 no lists,
# 0x400000 no flush).  tools/ab_ablate.sh [flags...]   (default: 0 1048576 3145728 4194304 7340032)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DRF_BRICK_ABLATE -I include thr3ed_atom_amd/csrc/relu_field_kernels.hip -o tools/exp_ablate.so || exit 1
[ $# -eq 0 ] && set -- 0 1048576 3145728 4194304 7340032
for f in "$@"; do
  RF_BRICK_STAGGER=$f RF_LIB_PATH=$PWD/tools/exp_ablate.so python bench.py --steps 40 --warmup 10 --render-frames 0 --highres-frames 0 --dropin-steps 0 --cpu-rays 0 2>/dev/null | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][0])
print('RF_BRICK_STAGGER=$f', 'brick pass %.4f ms' % l['kernels']['brick_accumulate']['avg_ms'])
"
done
