mkdir -p gpurun_out/r02d
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --timed-only $EXTRA > gpurun_out/r02d/$name.json 2> gpurun_out/r02d/$name.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02d/$name.json").read().strip().splitlines()[-1])
print("$name value %.3f G  ms/step %.2f" % (j["value"]/1e9, j["ms_per_step"]), {k:round(v,2) for k,v in j["kernel_ms_per_step"].items() if k!="note"}, {k:round(v,2) for k,v in j["host_ms_per_step"].items()})
PY
}
EXTRA="--pipe-mode 2 --steps 20" run m2
EXTRA="--pipe-mode 0 --steps 20" run m0
