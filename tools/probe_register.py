"""Probe: can caller-owned pageable memory be page-locked (hipHostRegister) on this box, and what does H2D run at
from pageable / registered / hipHostMalloc'd memory? Dev tool. (History: the library once registered the caller's
dirty span in place on the strength of these numbers; that was withdrawn — later pageable copies over the same
addresses aborted intermittently — and the span now travels through the library's own pinned chunks)"""
import ctypes as C, time, numpy as np, sys
hip = C.CDLL("libamdhip64.so")
n = 800 << 20
a = np.ones(n, np.uint8)
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), C.c_size_t(n)) == 0
def h2d(ptr, label):
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        assert hip.hipMemcpy(dev, C.c_void_p(ptr), C.c_size_t(n), 1) == 0
    hip.hipDeviceSynchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{label}: {n/dt/1e9:.1f} GB/s ({dt*1e3:.1f} ms for 800 MiB)")
h2d(a.ctypes.data, "pageable")
t0 = time.perf_counter()
rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(n), 0)
print("hipHostRegister rc", rc, f"{(time.perf_counter()-t0)*1e3:.1f} ms")
if rc == 0:
    h2d(a.ctypes.data, "registered")
    t0 = time.perf_counter(); hip.hipHostUnregister(C.c_void_p(a.ctypes.data)); print(f"unregister {(time.perf_counter()-t0)*1e3:.1f} ms")
p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), C.c_size_t(n), 0) == 0
C.memset(p, 1, n)
h2d(p.value, "hipHostMalloc")
