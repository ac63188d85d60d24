"""Measures the PCIe-inclusive rate of the host-buffer entry point lsq_encode_icm (what a Julia caller uses)
through the plain-C consumer (tests/c_abi_consumer.c): cfg2, 10^6 x 128 f32, m = 8, 16 ILS x 4 sweeps."""
import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from test_gpu_parity import _run_c_consumer
n, d, m = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 128, 8
X = O.synth_data_u8(1234, n, d)
K = np.ascontiguousarray(O.synth_data_u8(4321, m * 256, d) / np.float32(m))
B0 = O.randinit(7, n, m, 256)
with tempfile.TemporaryDirectory() as t:
    Bs, objs, secs, out = _run_c_consumer(pathlib.Path(t), X, B0, K, m, [16], 4, 4, True, 42)
print(out.strip())
print("host-buffer (PCIe-inclusive) rate: %.0f vectors/s, %.1f ms per call" % (n / secs, secs * 1e3))
