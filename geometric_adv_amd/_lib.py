"""ctypes binding of libgeoadv.so (include/geoadv.h).  There is NO fallback: if the HIP library
is missing or a call fails, this raises -- the product never routes through a CPU path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgeoadv.so")

_lib = None


class GeoAdvError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GeoAdvError(
                "libgeoadv.so not built: %s is missing. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C geometric_adv_amd/csrc` (needs hipcc, --offload-arch=gfx950)." % LIB_PATH)
        import torch  # noqa: F401  -- first: libgeoadv.so must bind to the HIP runtime torch ships, not to a second copy
        _lib = C.CDLL(LIB_PATH)
        _lib.geoadv_last_error.restype = C.c_char_p
        for name in ("geoadv_approx_match_temp_floats", "geoadv_ae_workspace_bytes", "geoadv_chamfer_matrix_workspace_floats",
                     "geoadv_emd_cost_grad1_temp_floats", "geoadv_nn_distance_sym_workspace_floats",
                     "geoadv_knn_workspace_bytes", "geoadv_group_point_grad_workspace_bytes", "geoadv_match_cost_workspace_floats"):
            getattr(_lib, name).restype = C.c_size_t
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().geoadv_last_error().decode("utf-8", "replace")
        if status == 1:
            raise ValueError("%s: %s" % (what, msg))
        raise GeoAdvError("%s failed (status %d): %s" % (what, status, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_handle():
    """hipStream_t of torch's current stream, as void*."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
