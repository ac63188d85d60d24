"""The reference's demos/demo_lsq_gpu.jl flow against this package (BASELINE's secondary metric, recall@1 on SIFT1M):
OPQ init -> ChainQ init -> train_lsq -> encode the base set on the GPU -> quantise norms -> ADC linear scan -> recall.

    LSQ_DATA_DIR=/data python tools/demo_lsq_gpu.py [nread_train] [nread_base] [nquery]

needs $LSQ_DATA_DIR/sift/{sift_learn,sift_base,sift_query}.fvecs and sift_groundtruth.ivecs (TEXMEX layout).  The data is
not in the build image; without it the script says so and runs the same flow on a small synthetic stand-in, so that the
wiring stays exercised (tests/test_pipeline_gpu.py asserts on that flow)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lsq = importlib.import_module("local-search-quantization_amd")


def load(nt, nb, nq):
    base = os.path.join(os.environ.get("LSQ_DATA_DIR", ""), "sift")
    names = ["sift_learn.fvecs", "sift_base.fvecs", "sift_query.fvecs", "sift_groundtruth.ivecs"]
    if os.environ.get("LSQ_DATA_DIR") and all(os.path.exists(os.path.join(base, f)) for f in names):
        xt = lsq.fvecs_read(nt, os.path.join(base, names[0]))
        xb = lsq.fvecs_read(nb, os.path.join(base, names[1]))
        xq = lsq.fvecs_read(nq, os.path.join(base, names[2]))
        gt = lsq.ivecs_read(nq, os.path.join(base, names[3]))[0] + 1          # 0-based in the file (demo_lsq_gpu.jl:62-64)
        if nb < 1_000_000:                                                    # ground truth of a prefix: recompute exactly
            gt = (((xb[:, :, None] - xq[:, None, :]) ** 2).sum(0)).argmin(0) + 1 if nb * nq <= 4e8 else gt
        return "SIFT1M", xt, xb, xq, gt.astype(np.uint32)
    print("SIFT1M not found under $LSQ_DATA_DIR/sift -- running the synthetic stand-in (clustered Gaussians, d = 32)")
    rng = np.random.default_rng(1)
    d, k = 32, 400
    cen = rng.standard_normal((d, k)).astype(np.float32) * 3.0
    allx = (cen[:, rng.integers(k, size=nt + nb + nq)] + 0.35 * rng.standard_normal((d, nt + nb + nq))).astype(np.float32)
    xt, xb, xq = allx[:, :nt], allx[:, nt:nt + nb], allx[:, nt + nb:]
    gt = (((xb[:, :, None] - xq[:, None, :]) ** 2).sum(0)).argmin(0) + 1
    return "synthetic", xt, xb, xq, gt.astype(np.uint32)


def main():
    real = bool(os.environ.get("LSQ_DATA_DIR"))
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else (10_000 if real else 3000)
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else (1_000_000 if real else 6000)
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else (10_000 if real else 64)
    name, x_train, x_base, x_query, gt = load(nt, nb, nq)
    d = x_train.shape[0]
    m, h, niter, knn = (7, 256, 10, 1000) if name == "SIFT1M" else (4, 256, 3, 50)     # demo_lsq_gpu.jl:13-20
    C, B, R, err = lsq.train_opq(x_train, m, h, niter, "natural", True)
    print("Error after OPQ is %e" % err[-1])
    C, B, R, err = lsq.train_chainq(x_train, m, h, R, B, C, niter)
    print("Error after ChainQ is %e" % err[-1])
    ilsiter, icmiter, randord, npert = 8, 4, True, 4
    C, B, cbnorms, B_norms, obj = lsq.train_lsq(x_train, m, h, R, B, C, niter, ilsiter, icmiter, randord, min(npert, m), True)
    B_base = lsq.randinit(x_base.shape[1], m, h)
    t0 = time.perf_counter()
    Bs, objs = lsq.encode_icm_cuda(x_base, B_base, C, [16], icmiter, min(npert, m), randord, 2, True)
    dt = time.perf_counter() - t0
    B_base = Bs[-1]
    print("Encoded %d base vectors in %.3f s (%.0f vectors/s, host buffers); error in base is %e" % (x_base.shape[1], dt, x_base.shape[1] / dt, objs[-1]))
    with lsq.Engine(0) as eng:                                              # norm quantisation on the device (lsq_quantize_norms), checked against the mirror
        nidx = lsq.quantize_norms(B_base, C, cbnorms, engine=eng)
    assert np.array_equal(nidx, lsq.quantize_norms(B_base, C, cbnorms)), "device and host norm quantisation disagree"
    db_norms = np.asarray(cbnorms, dtype=np.float32)[nidx.astype(np.int64) - 1]
    t0 = time.perf_counter()
    with lsq.Engine(0) as eng:                                              # the ADC scan on the device (lsq_linscan) ...
        dists, idx = lsq.linscan_lsq((B_base - 1).astype(np.uint8), x_query, C, db_norms, np.eye(d, dtype=np.float32), knn, engine=eng)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    dists_h, idx_h = lsq.linscan_lsq((B_base - 1).astype(np.uint8), x_query, C, db_norms, np.eye(d, dtype=np.float32), knn)   # ... and on the host cores
    t_host = time.perf_counter() - t0
    assert np.array_equal(idx, idx_h) and np.array_equal(dists, dists_h), "device and host scans disagree"
    print("Searched %d queries: device %.3f s (host buffers, incl. copies), host %.3f s; identical results" % (x_query.shape[1], t_dev, t_host))
    rec = lsq.eval_recall(gt, idx.astype(np.uint32), knn, True)
    print("%s: recall@1 = %.4f, recall@%d = %.4f" % (name, rec[0], knn, rec[knn - 1]))


if __name__ == "__main__":
    main()
