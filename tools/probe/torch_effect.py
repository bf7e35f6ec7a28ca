"""Probe: what kind of host work gets slower after `import torch`?  python tools/probe/torch_effect.py [torch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "torch" in sys.argv:
    import torch
import ctypes
import numpy as np
from limap_amd import _capi
hip = ctypes.CDLL("libamdhip64.so")  # whichever runtime _capi bound
def med(f, n=200):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort(); return 1e6 * ts[len(ts) // 2]
a = np.random.rand(200_000); b = np.empty_like(a)          # 1.6 MB
big = np.random.rand(8_000_000); bigb = np.empty_like(big)  # 64 MB
def pyloop():
    s = 0
    for i in range(20000): s += i
    return s
stream = ctypes.c_void_p(); hip.hipStreamCreate(ctypes.byref(stream))
ev = ctypes.c_void_p(); hip.hipEventCreate(ctypes.byref(ev))
d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), 1 << 24)
h = ctypes.c_void_p(); hip.hipHostMalloc(ctypes.byref(h), 1 << 24, 0)
def h2d():
    hip.hipMemcpyAsync(d, h, 1 << 20, 1, stream); hip.hipStreamSynchronize(stream)
def h2d_pageable():
    hip.hipMemcpyAsync(d, ctypes.c_void_p(a.ctypes.data), 1 << 20, 1, stream); hip.hipStreamSynchronize(stream)
def evrec():
    hip.hipEventRecord(ev, stream)
def alloc():
    x = np.zeros(1_000_000); x[::512] = 1.0
h2d(); h2d_pageable()
print(("torch " if "torch" in sys.argv else "plain ") +
      "copy1.6MB %.1f us  copy64MB %.0f us  pyloop %.0f us  h2d_1MB_pinned %.1f us  h2d_1MB_pageable %.1f us  eventRecord %.2f us  zeros8MB+touch %.0f us  cpu %d" % (
      med(lambda: np.copyto(b, a)), med(lambda: np.copyto(bigb, big), 20), med(pyloop, 50), med(h2d), med(h2d_pageable), med(evrec, 2000), med(alloc, 50),
      int(open("/proc/self/stat").read().split()[38])))
