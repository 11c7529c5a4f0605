# Read-only probe (run on the GPU box): can this MI355X be seen as several logical devices (compute partitions), so that the N > 1
# RCCL paths could run on one box?  Writes gpurun_out/partitions/probe.txt.  Changes nothing.
OUT=gpurun_out/partitions
mkdir -p $OUT
{
  echo "== devices"; ls -l /dev/kfd /dev/dri 2>&1
  echo "== rocm-smi partitions"; timeout 60 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -40
  echo "== sysfs"
  for d in /sys/class/drm/card*/device; do
    for f in current_compute_partition available_compute_partition current_memory_partition available_memory_partition; do
      [ -e $d/$f ] && { echo "$d/$f: $(cat $d/$f 2>&1)  [$(stat -c '%A %U' $d/$f)]"; }
    done
  done
  echo "== writable?"; for d in /sys/class/drm/card*/device; do [ -w $d/current_compute_partition ] && echo "$d/current_compute_partition writable" || echo "$d: not writable"; done
  mount | grep -E " /sys " | head
  echo "== rocminfo agents"; timeout 60 rocminfo 2>&1 | grep -E "Marketing Name|Compute Unit|Node:|Uuid" | head -40
  echo "== torch"; python -c "import torch; print(torch.cuda.device_count(), [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())])"
  echo "== env"; env | grep -E "HIP_VISIBLE|ROCR_VISIBLE|CUDA_VISIBLE|GPU_DEVICE"
} > $OUT/probe.txt 2>&1
cat $OUT/probe.txt
