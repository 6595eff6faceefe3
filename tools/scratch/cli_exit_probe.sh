#!/bin/bash
# Developer probe (GPU box): how much of a fresh command-line process is tear-down after main() returned.
D=/dev/shm/sd_exit_probe; mkdir -p $D
python3 -c "
import sys, os
sys.path.insert(0, os.getcwd())
from stringdecomposer_amd import synth
mn, ms = synth.make_monomers(12, seed=1)
rn, rs = synth.make_reads(ms, 1000, read_len=50000, seed=1)
synth.write_fasta('$D/r.fa', rn, rs, width=80); synth.write_fasta('$D/m.fa', mn, ms)
"
cat > $D/run.py <<PY
import sys, os, time, pathlib
t0 = time.perf_counter()
sys.path.insert(0, os.getcwd())
import stringdecomposer_amd
stringdecomposer_amd.prefer_queue_thread_dispatch()
import stringdecomposer_amd.main as cli
t1 = time.perf_counter()
sys.argv = ["stringdecomposer", "$D/r.fa", "$D/m.fa", "-o", "$D/out", "-t", "32"]
rc = cli.main()
t2 = time.perf_counter()
sys.stderr.write("imports %.0f ms, main() %.0f ms\n" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
sys.stderr.flush()
if os.environ.get("FAST_EXIT"):
    os._exit(0)
PY
python3 - <<PY
import subprocess, time, os, sys
for env_extra in [{}, {}, {}, {"SD_MIN_BATCHES": "1", "SD_CLEAN_EXIT": "1"}, {"SD_MIN_BATCHES": "1", "SD_CLEAN_EXIT": "1"}, {}, {"SD_MIN_BATCHES": "1", "SD_CLEAN_EXIT": "1"}, {}, {}]:
    env = dict(os.environ, SD_TIMING="1", **env_extra)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "bin/stringdecomposer", "$D/r.fa", "$D/m.fa", "-o", "$D/out", "-t", "32"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    dt = time.perf_counter() - t0
    tl = [l for l in p.stdout.decode().splitlines() if "batches:" in l or "engine (" in l or "allocations" in l]
    print("%-34s rc %d wall %.0f ms | %s" % ("one batch, ordinary exit (round 3)" if env_extra else "default", p.returncode, dt * 1e3, " | ".join(x[12:].strip()[:110] for x in tl)), flush=True)
PY
sha256sum $D/out/*.tsv | cut -c1-16
rm -rf $D
