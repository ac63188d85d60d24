"""tools/first_call.py [n] [d] -- run ON THE GPU BOX in a FRESH process: what the FIRST host-buffer call of a process costs (the Julia demo makes one call per
process: demos/demo_lsq_gpu.jl:50) next to the calls that follow, and where the difference goes (context creation, workspace allocation, first-touch)."""
import ctypes as C, importlib, sys, time, numpy as np
sys.path.insert(0, ".")
t00 = time.perf_counter()
import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reserve = int(sys.argv[3]) if len(sys.argv) > 3 else 0
m = 8
rng = np.random.default_rng(0)
Xh = rng.integers(0, 256, size=(n, d)).astype(np.float32)          # touched host pages (any caller's data are)
Kh = (rng.integers(0, 256, size=(m * 256, d)) / 8.0).astype(np.float32)
Bh = rng.integers(1, 257, size=(n, m)).astype(np.int16)
lsq = importlib.import_module("local-search-quantization_amd")
t0 = time.perf_counter()
eng = lsq.Engine(0, profile=bool(int(sys.argv[4])) if len(sys.argv) > 4 else False)
t1 = time.perf_counter()
print("lsq_create: %.1f ms" % ((t1 - t0) * 1e3))
if reserve:
    t0 = time.perf_counter(); eng.reserve(n, d, m); print("lsq_reserve: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for call in range(4):
    t0 = time.perf_counter()
    Bs, objs = eng.encode_icm(Xh, Bh, Kh, m, [16], 4, 4, True, seed=42)
    dt = time.perf_counter() - t0
    tm = eng.timings(); eng.reset_timings()
    print("call %d: %.2f ms  (%.2f M vectors/s)" % (call, dt * 1e3, n / dt / 1e6), {k: round(v, 2) for k, v in tm.items() if isinstance(v, float) and v}, flush=True)
eng.close()
# raw allocator cost of the call's large buffers, for scale
hip = C.CDLL("libamdhip64.so")
for gb in (0.5, 4.0, 8.0):
    p = C.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(int(gb * (1 << 30)))); t1 = time.perf_counter()
    hip.hipFree(p)
    print("hipMalloc %.1f GiB: %.2f ms (rc %d), hipFree %.2f ms" % (gb, (t1 - t0) * 1e3, rc, (time.perf_counter() - t1) * 1e3))
