mkdir -p gpurun_out
for s in 5004 5005 5006 5007 5008 5009; do timeout 600 python tools/fuzz_gpu.py 1500 $s gpurun_out/r03_fuzz_c.txt 2>&1 | tail -1; done
for s in 64 65 66 67; do timeout 400 python tools/fuzz_final.py 400 $s gpurun_out/r03_fuzz_final_c.txt 2>&1 | tail -1; done
