#!/usr/bin/env python3
"""Developer tool (needs a GPU): CPU time per host thread and pipelined C2 step -- which threads of a rank are
busy while the device works (host workers, the waiting main thread, runtime helper threads).
usage: python tools/thread_cpu_probe.py [host threads] [block]   (block: hipDeviceScheduleBlockingSync first)"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from stringdecomposer_amd import lib, synth
def snap():
    out={}
    for t in os.listdir('/proc/self/task'):
        try:
            f=open('/proc/self/task/%s/stat'%t).read()
            comm=f[f.index('(')+1:f.rindex(')')]
            rest=f[f.rindex(')')+2:].split()
            out[t]=(comm,(int(rest[11])+int(rest[12]))*10.0)  # ms at 100 Hz
        except Exception: pass
    return out
mn, ms = synth.make_monomers(12, seed=1)
rn, rs = synth.make_reads(ms, 1000, read_len=50000, seed=1)
rset = lib.ReadSet(rs)
thr=int(sys.argv[1]) if len(sys.argv)>1 else 16
if len(sys.argv)>2 and sys.argv[2]=="block":
    import ctypes
    hip=ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(BlockingSync) ->", hip.hipSetDeviceFlags(4))
st = lib.Stream(ms, device=0, threads=thr)
for _ in range(3):
    st.submit(rset); st.collect()
a=snap(); t0=time.perf_counter()
K=40
st.submit(rset)
for k in range(K-1):
    st.submit(rset); st.collect()
st.collect()
dt=time.perf_counter()-t0; b=snap()
d=sorted(((b[t][1]-a.get(t,(0,0))[1], b[t][0], t) for t in b), reverse=True)
print("threads",thr,"ms/step", dt/K*1e3, "cpu ms/step total", sum(x[0] for x in d)/K)
for x in d[:8]: print("   %-16s tid %s  %.2f ms/step" % (x[1], x[2], x[0]/K))
