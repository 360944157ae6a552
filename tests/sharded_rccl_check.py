"""Child process of test_gpu_sharded_rccl.py: drives sift_hip_match_u8_sharded with a
one-rank RCCL communicator created through ctypes (no torch in this process, so the
library's run-time lookup of ncclAllGather and the communicator come from one RCCL)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sift_amd  # noqa: E402


def load(names):
    for n in names:
        try:
            return C.CDLL(n, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise SystemExit("cannot load any of " + ", ".join(names))


rccl = load(["librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"])
hip = load(["libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"])
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]
H2D, D2H = 1, 2


def dev_from(a):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), max(a.nbytes, 4)) == 0
    assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, H2D) == 0
    return p


ctx = sift_amd.SiftHip(0)
comm = C.c_void_p()
devs = (C.c_int * 1)(0)
assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
nq, nr = 700, 5000
q, r = sift_amd.planted_descriptors(nq, nr, plant=0.05)
dq, dr = dev_from(q), dev_from(r)
outs = [np.zeros(nq, np.int32) for _ in range(3)]
douts = [dev_from(o) for o in outs]
rc = ctx.lib.sift_hip_match_u8_sharded(ctx.h, comm, 1, dq, nq, dr, nr, 0.5, 0, douts[0], douts[1], douts[2])
assert rc == 0, (rc, ctx.lib.sift_hip_last_error(ctx.h))
ctx.synchronize()
for o, d in zip(outs, douts):
    assert hip.hipMemcpy(o.ctypes.data_as(C.c_void_p), d, o.nbytes, D2H) == 0
best, d1, d2 = ctx.match_u8(q, r, 0.5, 0)
assert np.array_equal(outs[0], best) and np.array_equal(outs[1], d1) and np.array_equal(outs[2], d2)
assert (best >= 0).sum() >= 20
# the ragged form (per-rank counts on the host, padded gather + compaction) on the same communicator
counts = np.array([nr], np.int32)
for o, d in zip(outs, douts):
    o[:] = -5
    assert hip.hipMemcpy(d, o.ctypes.data_as(C.c_void_p), o.nbytes, H2D) == 0
rc = ctx.lib.sift_hip_match_u8_sharded_ragged(ctx.h, comm, 1, 0, dq, nq, dr, counts.ctypes.data_as(C.c_void_p), 0.5, 0,
                                              douts[0], douts[1], douts[2])
assert rc == 0, (rc, ctx.lib.sift_hip_last_error(ctx.h))
ctx.synchronize()
for o, d in zip(outs, douts):
    assert hip.hipMemcpy(o.ctypes.data_as(C.c_void_p), d, o.nbytes, D2H) == 0
assert np.array_equal(outs[0], best) and np.array_equal(outs[1], d1) and np.array_equal(outs[2], d2)
bad = np.array([-1], np.int32)
assert ctx.lib.sift_hip_match_u8_sharded_ragged(ctx.h, comm, 1, 0, dq, nq, dr, bad.ctypes.data_as(C.c_void_p), 0.5, 0,
                                                douts[0], douts[1], douts[2]) == sift_amd.capi.ERR_ARG
rccl.ncclCommDestroy(comm)
for d in [dq, dr] + douts:
    hip.hipFree(d)
print("sharded-rccl ok: accepted", int((best >= 0).sum()))
