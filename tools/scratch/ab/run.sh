# A/B of two builds of libsd_hip.so on one box: the tree's library against tools/scratch/ab/libsd_hip_<tag>.so (SD_HIP_LIB)
T=${1:-w5}
for i in 1 2 3; do
  for v in base $T; do
    if [ $v = base ]; then unset SD_HIP_LIB; else export SD_HIP_LIB=$PWD/tools/scratch/ab/libsd_hip_$v.so; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(j['ms_per_step'],3), 'sustained', round(j['sustained']['ms_per_step'],3), 'alone fill', round(j['device_resident']['kernel_ms_per_step']['fill'],3), j['rows_out_per_gpu'])"
  done
done
