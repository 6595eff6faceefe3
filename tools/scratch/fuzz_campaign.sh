mkdir -p gpurun_out/fz
for s in 9901 9902 9903 9904 9905 9906 9907 9908; do timeout 400 python tools/fuzz_gpu.py 500 $s gpurun_out/fz/fuzz.txt > gpurun_out/fz/f$s.log 2>&1; tail -1 gpurun_out/fz/f$s.log; done
for s in 9911 9912 9913; do timeout 300 python tools/fuzz_final.py 120 $s gpurun_out/fz/fuzz_final.txt > gpurun_out/fz/ff$s.log 2>&1; tail -1 gpurun_out/fz/ff$s.log; done
for s in 31 32 33 34; do timeout 300 python tools/fuzz_stream.py 50 $s gpurun_out/fz/fuzz_stream.txt > gpurun_out/fz/fs$s.log 2>&1; tail -1 gpurun_out/fz/fs$s.log; done
