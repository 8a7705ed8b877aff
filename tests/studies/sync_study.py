"""How quickly a PML walk started in the middle of a read falls into step with the walk of the whole read
(same row, offset and match length -- from then on the two are identical for good).  The data behind the
"segment-parallel long reads" item of DESIGN.md: test infrastructure only (uses the oracle)."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.oracle import Oracle, lib

d = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.08
n = int(sys.argv[4]) if len(sys.argv) > 4 else 300
tool = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools", "build_index")
rf = "/tmp/sync_reads_%d_%g.bin" % (L, sub)
subprocess.check_call([tool, "reads", os.path.join(d, "text.bin"), str(n), str(L), str(sub), "5", rf], stderr=subprocess.DEVNULL)
reads = np.fromfile(rf, np.uint8).reshape(n, L)
img = open(os.path.join(d, "index.movi"), "rb").read()
ora = Oracle(img)
Lb = lib()
Lb.oracle_pml_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]


def trace(r):
    r = np.ascontiguousarray(r)
    out = np.zeros(r.size, np.uint16); idx = np.zeros(r.size, np.uint64); off = np.zeros(r.size, np.uint32)
    assert Lb.oracle_pml_trace(ora._h, r.ctypes.data, r.size, out.ctypes.data, idx.ctypes.data, off.ctypes.data) == 0
    return out, idx, off


dist = []
never = 0
for i in range(n):
    r = reads[i]
    t_out, t_idx, t_off = trace(r)                      # emission order: step k = position L-1-k
    for e in (L // 4, L // 2, 3 * L // 4):              # a walk started fresh at position e-1 (steps of the prefix r[:e])
        s_out, s_idx, s_off = trace(r[:e])
        k0 = L - e                                       # step of the full walk that handles position e-1
        same = (s_out == t_out[k0:]) & (s_idx == t_idx[k0:]) & (s_off == t_off[k0:])
        w = np.flatnonzero(same)
        if w.size == 0:
            never += 1
            continue
        first = int(w[0])
        assert same[first:].all()                       # once in step, in step for good
        dist.append(first)
dist = np.array(dist)
print("reads %d x %d bp, %.1f %% substitutions: %d walks started mid-read; never in step before the read ended: %d" % (n, L, 100 * sub, dist.size + never, never))
print("bases until in step: median %d, mean %.1f, 90 %% %d, 99 %% %d, max %d" % (np.median(dist), dist.mean(), np.percentile(dist, 90), np.percentile(dist, 99), dist.max()))
