"""stringdecomposer_amd -- MI355X-native StringDecomposer read x monomer DP hot path."""
import os as _os
import sys as _sys

__version__ = "0.1.0"


def prefer_queue_thread_dispatch():
    """Opt-in process setting for the entry points that OWN their process (bin/stringdecomposer, bench.py, tools/):
    AMD_DIRECT_DISPATCH=0 unless the caller has chosen.

    The HIP runtime's "direct dispatch" mode runs a helper thread per process that stays busy while kernels are in
    flight: 12.5 ms of CPU per 16.4-ms pipelined C2 step, measured (tools/helper_thread_ab.sh,
    profiles/r03_helper_thread.txt) -- two thirds of everything a rank asks of the host.  With the runtime's
    queue-thread mode the same step costs 12 ms of CPU instead of 17 at the same step time, which is what lets 8 ranks
    fit the 16-CPU quota of a GPU box.  The runtime reads the variable when it initialises, so this must run before
    the first HIP call of the process; importing the package does NOT change the environment (a library must not
    reconfigure the runtime of torch / RCCL users that merely import it).  Returns the effective value; warns when a
    HIP runtime is already loaded and the variable was not set (then it may be too late)."""
    if "AMD_DIRECT_DISPATCH" not in _os.environ:
        late = any(m in _sys.modules for m in ("torch",)) and getattr(_sys.modules.get("torch"), "cuda", None) is not None \
            and _sys.modules["torch"].cuda.is_initialized()
        if late:
            import warnings
            warnings.warn("AMD_DIRECT_DISPATCH chosen after the HIP runtime initialised: it may have no effect")
        _os.environ["AMD_DIRECT_DISPATCH"] = "0"
    return _os.environ["AMD_DIRECT_DISPATCH"]


def leave_without_teardown(rc):
    """End the process without the interpreter's and the HIP runtime's tear-down (0.1 s of a 0.3-s job: streams,
    events, 4+ GB of device buffers handed back one by one) -- for entry points that OWN their process
    (bin/stringdecomposer).  Nothing that was written is lost, whether or not the caller closed it: the exit handlers
    run (coverage, tracing tools), every Python file object that is still open is flushed (found through the collector,
    so a file a caller forgot to close is covered too), logging is shut down, and the C stdio buffers of every native
    library are flushed; bytes handed to write()/pwrite() or stored through a shared mapping are the kernel's already.
    The process's memory goes back to the driver either way."""
    import atexit
    import ctypes
    import gc
    import io
    import logging
    atexit._run_exitfuncs()
    logging.shutdown()
    for o in gc.get_objects():
        try:
            if isinstance(o, io.IOBase) and not o.closed and o.writable():
                o.flush()
        except Exception:                # a broken pipe on stdout must not turn into a traceback here
            pass
    ctypes.CDLL(None).fflush(None)
    _os._exit(rc if isinstance(rc, int) else (0 if rc is None else 1))
