"""ctypes loader of tools/probe/libgeoadv_probe.so (calibration kernels; built by `make -C tools/probe`)."""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libgeoadv_probe.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE], check=True)
        _LIB = C.CDLL(path)
        _LIB.geoadv_probe_microbench.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
        _LIB.geoadv_probe_last_error.restype = C.c_char_p
    return _LIB


def microbench(which, iters=2000):
    """ms for 2048 x 256 threads x 16 * iters VALU instructions of kind `which` (see microbench.hip)."""
    ms = C.c_float(0)
    st = lib().geoadv_probe_microbench(int(which), int(iters), C.byref(ms), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if st != 0:
        raise RuntimeError("probe microbench failed: %s" % lib().geoadv_probe_last_error().decode())
    return ms.value
