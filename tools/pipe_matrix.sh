mkdir -p gpurun_out/r02d
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline $EXTRA > gpurun_out/r02d/$name.json 2> gpurun_out/r02d/$name.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02d/$name.json").read().strip().splitlines()[-1])
print("$name value %.3f G  ms/step %.2f" % (j["value"]/1e9, j["ms_per_step"]), {k:round(v,2) for k,v in j["kernel_ms_per_step"].items() if k!="note"}, {k:round(v,2) for k,v in j["host_ms_per_step"].items()}, "resident %.3f G" % (j["device_resident"]["bp_per_s"]/1e9), {k:round(v,2) for k,v in j["device_resident"]["kernel_ms_per_step"].items()})
PY
}
run m0 SD_PIPE_MODE=0
run m1_prio1 SD_PIPE_MODE=1
run m1_prio0 SD_PIPE_MODE=1 SD_PIPE_PRIO=0
EXTRA="--sub-batches 2" run m0_sb2 SD_PIPE_MODE=0
EXTRA="--sub-batches 2" run m1_sb2 SD_PIPE_MODE=1
EXTRA="--steps 30" run m1_k30 SD_PIPE_MODE=1
EXTRA="--steps 30" run m0_k30 SD_PIPE_MODE=0
