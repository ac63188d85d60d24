import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsq_amd as L, oracle as O
eng = L.Engine(0)
for (n, d) in [(10, 1030), (10, 1024), (3, 2048), (10, 130)]:
    X2 = eng.synth_data_u8_dev(5, n, d).cpu().numpy()
    R = O.synth_data_u8(5, n, d)
    bad = np.argwhere(X2 != R)
    print(n, d, "mismatches", len(bad), bad[:8].tolist(), X2[tuple(bad[0])] if len(bad) else None, R[tuple(bad[0])] if len(bad) else None)
