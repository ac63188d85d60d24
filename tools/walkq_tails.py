"""tools/walkq_tails.py [n] -- run ON THE GPU BOX with the tuning library.  Start / end clock of every block of the LAST icm_walkq_kernel launch of an
encode (production schedule): how much of a launch is tail (CUs idle while the slowest block finishes)."""
import ctypes as C, importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
lsq = importlib.import_module("local-search-quantization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = lsq._lib.load(tuning=True)
buf = torch.zeros((512, 2), dtype=torch.int64, device="cuda")
L.lsq_tuning_set_walkq_block_clock.restype = C.c_int
L.lsq_tuning_set_walkq_block_clock.argtypes = [C.c_void_p]
assert L.lsq_tuning_set_walkq_block_clock(buf.data_ptr()) == 0
with lsq.Engine(0, schedule=6, tuning=True) as eng:
    dX = eng.synth_data_u8_dev(1234, n, 128); dB0 = eng.randinit_dev(7, n, 8); dK = eng.synth_codebooks_dev(4321, 8, 128)
    for ils in (1, 4, 16):
        buf.zero_()
        eng.encode_icm_dev(dX, dB0, dK, 8, [ils], 4, 4, True, seed=42)
        torch.cuda.synchronize()
        t = buf.cpu().numpy()
        t = t[t[:, 0] != 0]
        start, end = t[:, 0].min(), t[:, 1].max()
        busy = (t[:, 1] - t[:, 0]) / 100.0
        print("launch of ILS iteration %2d: %d blocks, launch %.1f us, block busy mean %.1f min %.1f max %.1f us, start spread %.1f us, mean idle at the end %.1f us (%.1f %% of the launch)" %
              (ils, len(t), (end - start) / 100.0, busy.mean(), busy.min(), busy.max(), (t[:, 0].max() - start) / 100.0, ((end - t[:, 1]) / 100.0).mean(),
               100.0 * ((end - t[:, 1]).mean()) / (end - start)))
