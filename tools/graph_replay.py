"""tools/graph_replay.py [n] [d] [m] [ils] -- run ON THE GPU BOX.  The device-resident encode (option "async": no host synchronisation inside the call) captured into a HIP
graph on a side stream and replayed, against the same call enqueued kernel by kernel: what the launch gaps between the ~40 dependent kernels of a call cost."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")
n, d, m, ils = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 1_000_000), (2, 128), (3, 8), (4, 16)))
with lsq.Engine(0) as eng:
    dX = eng.synth_data_u8_dev(1234, n, d)
    if d == 960:
        dX.mul_(0.3 / 255.0)
    dB0 = eng.randinit_dev(7, n, m)
    dK = eng.synth_codebooks_dev(4321, m, d)
    if d == 960:
        dK.mul_(0.3 / 255.0)
    out = torch.zeros((1, n, m), dtype=torch.uint8, device=dX.device)
    args = (dX, dB0, dK, m, [ils], 4, 4, True)
    ref, _, _ = eng.encode_icm_dev(*args, seed=42)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        eng.encode_icm_dev(*args, seed=42, out=out, nonblocking=True)
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            eng.encode_icm_dev(*args, seed=42, out=out, nonblocking=True)
    out.zero_(); graph.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(out[0], ref[0]))
    reps = 10
    for name, fn in (("stream (async call)", lambda: eng.encode_icm_dev(*args, seed=42, out=out, nonblocking=True)), ("graph replay", graph.replay),
                     ("stream (async call)", lambda: eng.encode_icm_dev(*args, seed=42, out=out, nonblocking=True)), ("graph replay", graph.replay)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-22s n=%d d=%d m=%d ils=%d: %.3f ms per call  (%.2f M vectors/s)  same codes as the blocking call: %s" % (name, n, d, m, ils, dt * 1e3, n / dt / 1e6, same), flush=True)
