export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02e; mkdir -p $O
(cd /tmp && SD_PIPE_MODE=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --sub-batches 1 > $O/pmc.log 2>&1)
(cd /tmp && SD_PIPE_MODE=0 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --sub-batches 1 > $O/trace.log 2>&1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/pmc/p_counter_collection.csv")))
print(rows[0].keys())
for r in rows:
    if "sd_fast" in r["Kernel_Name"]:
        print(r["Kernel_Name"][:30], r["Counter_Name"], r["Counter_Value"], r.get("Start_Timestamp"), r.get("End_Timestamp"))
tr=list(csv.DictReader(open("$O/trace/t_kernel_trace.csv")))
print(tr[0].keys())
for r in tr:
    print(r["Kernel_Name"][:30], r.get("Stream_Id"), r.get("Queue_Id"), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, int(r["Start_Timestamp"])/1e6)
PY
