"""Developer helper: what the per-call hipMalloc/hipFree of the host-pointer pipeline costs."""
import ctypes, time, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
sizes = [201326592, 134217728, 33554432, 33554432, 67108864, 4194304, 67108864, 16777216, 16777216]
for rep in range(3):
    t0 = time.perf_counter()
    ptrs = []
    for s in sizes:
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(s)) == 0
        ptrs.append(p)
    t1 = time.perf_counter()
    for p in ptrs:
        assert hip.hipFree(p) == 0
    t2 = time.perf_counter()
    print(f"{len(sizes)} buffers, {sum(sizes) / 1e6:.0f} MB: hipMalloc {1e3 * (t1 - t0):.2f} ms, hipFree {1e3 * (t2 - t1):.2f} ms")
