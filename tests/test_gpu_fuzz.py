"""Fresh random parity cases in every driver-run GPU suite: fixed-seed slices of the three randomised campaigns
(tools/fuzz_gpu.py: template-set shapes up to 600 monomers, scorings around every cell-format switch, chunk sizes, N, --ed_thr,
host threads, device batch sizes; tools/fuzz_final.py: sd_run_files against the oracle, the Python convert_tsv and the
reference's edlib; tools/fuzz_stream.py: the streaming API at every pipeline depth) -- about a minute in all.  The seeds are
fixed so that a failure can be replayed (tools/replay_case.py on the gpurun_out/fuzz_fail_* directory the tool leaves); the
builder's long campaigns with other seeds are logged under profiles/r0N_fuzz*.txt."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(tool, cases, seed, oracle):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-3000:]
    assert "%d cases, 0 mismatches" % cases in out, out[-1000:]
    return out


@pytest.mark.parametrize("seed", [6001, 6002])
def test_fuzz_slice_raw_rows_vs_oracle(oracle, seed):
    _run("fuzz_gpu.py", 120, seed, oracle)


def test_fuzz_slice_final_tsv_vs_oracle_python_and_edlib(oracle):
    out = _run("fuzz_final.py", 50, 6101, oracle)
    assert "device identities == edlib" in out


def test_fuzz_slice_streaming_api(oracle):
    _run("fuzz_stream.py", 20, 6201, oracle)
