"""Does the x3 encoder forward survive a non-zero LDS base?  A one-wave kernel holding LDS on every CU runs on a side stream
while the forward runs on the main one (debug)."""
import ctypes as C, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from geometric_adv_amd import weights as W
from geometric_adv_amd.autoencoder import PointNetAE
lib = C.CDLL(os.path.join(HERE, "..", "probe", "libgeoadv_probe_bf16x3.so"))
lib.bf16x3_last_error.restype = C.c_char_p
n, B = 2048, 32
w = W.randomized_weights(n, seed=3)
pc = torch.rand(B, n, 3, device="cuda") - 0.5
side = torch.cuda.Stream()
for arith in ("f32", "bf16x3"):
    ae = PointNetAE(w, n, encoder_arith=arith)
    z0 = torch.as_tensor(ae.forward(pc)[0]).clone()
    torch.cuda.synchronize()
    for lds in (0, 8 << 10, 40 << 10, 70 << 10):
        bad = 0
        for rep in range(5):
            if lds:
                rc = lib.bf16x3_lds_hog(int(os.environ.get('HOG_BLOCKS', '256')), lds, 20000, C.c_void_p(side.cuda_stream))
                assert rc == 0, lib.bf16x3_last_error()
            import time
            t0 = time.perf_counter()
            z = torch.as_tensor(ae.forward(pc)[0]).clone()
            torch.cuda.current_stream().synchronize()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            bad += int(not torch.equal(z, z0))
        print(arith, "hog lds", lds, "mismatching runs", bad, "of 5", "forward ms %.3f, hog tail ms %.3f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
