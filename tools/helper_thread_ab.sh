#!/bin/bash
# Which runtime settings change the CPU time of the HIP runtime's helper thread(s) per pipelined C2 step?
# usage (GPU box): bash tools/helper_thread_ab.sh <outfile>
O=${1:-gpurun_out/helper_ab.txt}
mkdir -p $(dirname $O)
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/thread_cpu_probe.py 16 2>&1 | grep -v "amdgpu.ids" | tail -5 >> $O; }
: > $O
run A=default
run AMD_DIRECT_DISPATCH=0
run A=default
run AMD_DIRECT_DISPATCH=0
run AMD_DIRECT_DISPATCH=0 SD_PIPE_MODE=0
echo "== bench.py default / AMD_DIRECT_DISPATCH=0" >> $O
for e in A=1 AMD_DIRECT_DISPATCH=0; do env $e python bench.py --steps 10 --no-cpu-baseline --timed-only 2>/dev/null | python3 -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e', 'ms/step %.2f' % j['ms_per_step'], 'cpu ms/step %.1f' % j['host_ms_per_step']['process_cpu_ms'], j['kernel_ms_per_step'])" >> $O; done
cat $O
