#!/usr/bin/env python3
"""tools/alloc_time.py -- development aid: what hipMalloc / hipFree cost by size on the box (the plans of the panel family allocate and free
gigabytes of temporaries)."""
import ctypes, json, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipDeviceSynchronize()
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipFree(p)
for gb in (0.25, 0.5, 1, 2, 4, 8):
    n = int(gb * (1 << 30))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), n); t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(json.dumps({"GiB": gb, "rc": rc, "malloc_ms": [round(a, 2) for a, _ in ts], "free_ms": [round(b, 2) for _, b in ts]}))
