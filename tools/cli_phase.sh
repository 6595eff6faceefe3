#!/bin/bash
# Where a fresh command-line process spends its time (C4, --second-best): import times, the library's stage times
# (SD_TIMING), wall time.  usage (GPU box): bash tools/cli_phase.sh
D=/dev/shm/sd_cli_phase
mkdir -p $D
python3 -c "
import sys, os
sys.path.insert(0, os.getcwd())
from stringdecomposer_amd import synth
mn, ms = synth.make_monomers(64, seed=1)
rn, rs = synth.make_reads(ms, 256, read_len=50000, seed=1)
synth.write_fasta('$D/r.fa', rn, rs, width=80); synth.write_fasta('$D/m.fa', mn, ms)
"
for i in 1 2 3; do
  s=$(date +%s.%N)
  SD_TIMING=1 python3 -X importtime bin/stringdecomposer $D/r.fa $D/m.fa -o $D/out -t 32 --second-best 2> $D/err.txt
  e=$(date +%s.%N)
  echo "run $i: wall $(echo "$e - $s" | bc) s"
  grep "sd timing" $D/err.txt | cut -c1-200
  grep "import time" $D/err.txt | sort -t'|' -k2 -n | tail -3
done
rm -rf $D
