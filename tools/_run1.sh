set -x
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_c1.json 2> gpurun_out/r02_c1.err; tail -c 1500 gpurun_out/r02_c1.json; tail -5 gpurun_out/r02_c1.err
