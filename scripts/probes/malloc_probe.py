import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def malloc(n):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
    return p, rc, time.perf_counter() - t0
hip.hipSetDevice(0)
p, rc, t = malloc(1 << 20); print("first 1 MB (context init)", rc, round(t, 3))
for gb in (1, 4, 4, 4, 16, 40):
    p, rc, t = malloc(int(gb * (1 << 30)))
    t0 = time.perf_counter(); hip.hipFree(p); tf = time.perf_counter() - t0
    print(f"hipMalloc {gb} GiB: rc {rc} {t * 1e3:.1f} ms, hipFree {tf * 1e3:.1f} ms")
# host pinned
hp = ctypes.c_void_p()
t0 = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(hp), ctypes.c_size_t(1 << 30), 0); print("hipHostMalloc 1 GiB", rc, round((time.perf_counter() - t0) * 1e3, 1), "ms")
