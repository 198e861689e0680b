"""Does hipMemcpyAsync from pinned host memory return before the copy has run?  Host time of the call against copy size, on an idle stream
(hipHostMalloc default flags, the allocation arp_context uses for its staging block), plus the device time of the copy by events."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
H2D, D2H = 1, 2
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0  # hipStreamNonBlocking
for flags, name in ((0, "hipHostMallocDefault"), (0x2, "hipHostMallocMapped"), (0x40000000, "hipHostMallocNonCoherent")):
    host, dev = C.c_void_p(), C.c_void_p()
    n = 64 << 20
    if hip.hipHostMalloc(C.byref(host), n, flags) != 0:
        print(name, "allocation failed"); continue
    assert hip.hipMalloc(C.byref(dev), n) == 0
    C.memset(host, 1, n)
    for kind, kname in ((H2D, "H2D"), (D2H, "D2H")):
        for mb in (1, 4, 16, 52, 64):
            size = mb << 20
            best_call = best_all = 1e9
            for _ in range(5):
                hip.hipStreamSynchronize(st)
                t0 = time.perf_counter()
                a, b = (dev, host) if kind == H2D else (host, dev)
                assert hip.hipMemcpyAsync(a, b, size, kind, st) == 0
                t1 = time.perf_counter()
                hip.hipStreamSynchronize(st)
                t2 = time.perf_counter()
                best_call = min(best_call, t1 - t0); best_all = min(best_all, t2 - t0)
            print(f"{name:26s} {kname} {mb:3d} MB: call returns after {best_call * 1e6:8.1f} us, copy done after {best_all * 1e6:8.1f} us ({size / best_all / 1e9:5.1f} GB/s)", flush=True)

# What follows a large asynchronous copy on the same stream: does the NEXT call (a fill kernel, a small copy back) return at once?
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
host, dev, small = C.c_void_p(), C.c_void_p(), C.c_void_p()
assert hip.hipHostMalloc(C.byref(host), 64 << 20, 0) == 0 and hip.hipMalloc(C.byref(dev), 64 << 20) == 0 and hip.hipHostMalloc(C.byref(small), 4096, 0) == 0
for mb in (1, 16, 52):
    for _ in range(3):
        hip.hipStreamSynchronize(st)
        t0 = time.perf_counter()
        assert hip.hipMemcpyAsync(dev, host, mb << 20, H2D, st) == 0
        t1 = time.perf_counter()
        assert hip.hipMemsetAsync(dev, 0, 256, st) == 0
        t2 = time.perf_counter()
        assert hip.hipMemsetAsync(dev, 0, 256, st) == 0
        t3 = time.perf_counter()
        assert hip.hipMemcpyAsync(small, dev, 256, D2H, st) == 0
        t4 = time.perf_counter()
        hip.hipStreamSynchronize(st)
        t5 = time.perf_counter()
    print(f"{mb:3d} MB H2D then: copy call {1e6 * (t1 - t0):7.1f} us | first fill call {1e6 * (t2 - t1):7.1f} us | second fill call {1e6 * (t3 - t2):7.1f} us | small D2H call {1e6 * (t4 - t3):7.1f} us | drain {1e6 * (t5 - t4):7.1f} us", flush=True)
