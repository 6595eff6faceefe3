"""stringdecomposer_amd -- MI355X-native StringDecomposer read x monomer DP hot path."""
import os as _os

# The HIP runtime's "direct dispatch" mode runs a helper thread per process that stays busy while kernels are in
# flight: 12.5 ms of CPU per 16.4-ms pipelined C2 step, measured (tools/helper_thread_ab.sh, profiles/r03_helper_thread.txt)
# -- two thirds of everything a rank asks of the host.  With the runtime's queue-thread mode the same step costs 12 ms of
# CPU instead of 17 at the same step time, which is what lets 8 ranks fit the 16-CPU quota of a GPU box.  The runtime
# reads the variable when it initialises, so it is set here, at import, and only if the caller has not chosen.
_os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")

__version__ = "0.1.0"
