"""The reference's operator API for the attack path, on MI355X.

Function names, argument order and output order are those of the reference's Python wrappers
(external/structural_losses/tf_nndistance.py:15-41, tf_approxmatch.py:10-50,
external/grouping/tf_grouping.py:8-75).  Inputs and outputs are torch tensors on an AMD GPU
(containers only: the arithmetic runs in libgeoadv.so's hand-written gfx950 kernels).
Shape errors raise ValueError (the reference raises InvalidArgumentError from OP_REQUIRES).
"""
import ctypes as C

import torch

from . import _lib


def _f32(t, name, rank):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32 (got %s)" % (name, t.dtype))
    if t.dim() != rank:
        raise ValueError("%s must have rank %d (got shape %s)" % (name, rank, tuple(t.shape)))
    return t.contiguous()


def _i32(t, name, shape):
    if not t.is_cuda or t.dtype != torch.int32:
        raise ValueError("%s must be an int32 GPU tensor" % name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s must be of shape %s (got %s)" % (name, tuple(shape), tuple(t.shape)))
    return t.contiguous()


def _xyz_pair(xyz1, xyz2, op):
    xyz1 = _f32(xyz1, "xyz1", 3)
    xyz2 = _f32(xyz2, "xyz2", 3)
    if xyz1.shape[2] != 3:
        raise ValueError("%s only accepts 3d point set xyz1" % op)          # tf_nndistance.cpp:52
    if xyz2.shape[2] != 3:
        raise ValueError("%s only accepts 3d point set xyz2" % op)          # tf_nndistance.cpp:56
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("%s expects xyz1 and xyz2 have same batch size" % op)  # tf_nndistance.cpp:58
    return xyz1, xyz2


def nn_distance(xyz1, xyz2):
    """tf_nndistance.py:15-26.  xyz1 (b,n,3), xyz2 (b,m,3) ->
    dist1 (b,n) squared distance from each xyz1 point to its nearest xyz2 point, idx1 (b,n) int32,
    dist2 (b,m), idx2 (b,m).  Bit-identical to the reference CPU op; lowest index wins ties."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "NnDistance")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist1 = torch.empty((b, n), dtype=torch.float32, device=xyz1.device)
    idx1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    dist2 = torch.empty((b, m), dtype=torch.float32, device=xyz1.device)
    idx2 = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        st = _lib.lib().geoadv_nn_distance(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(dist1), _lib.ptr(idx1),
                                           _lib.ptr(dist2), _lib.ptr(idx2), _lib.stream_handle())
    _lib.check(st, "nn_distance")
    return dist1, idx1, dist2, idx2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """The NnDistanceGrad op behind tf_nndistance.py:35-41 -> (grad_xyz1, grad_xyz2)."""
    xyz1, xyz2 = _xyz_pair(xyz1, xyz2, "NnDistanceGrad")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    grad_dist1 = _f32(grad_dist1, "grad_dist1", 2)
    grad_dist2 = _f32(grad_dist2, "grad_dist2", 2)
    if tuple(grad_dist1.shape) != (b, n):
        raise ValueError("NnDistanceGrad requires grad_dist1 be of shape(batch,#points)")   # tf_nndistance.cpp:102
    if tuple(grad_dist2.shape) != (b, m):
        raise ValueError("NnDistanceGrad requires grad_dist2 be of shape(batch,#points)")   # tf_nndistance.cpp:104
    idx1 = _i32(idx1, "idx1", (b, n))
    idx2 = _i32(idx2, "idx2", (b, m))
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    with torch.cuda.device(xyz1.device):
        st = _lib.lib().geoadv_nn_distance_grad(b, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2), _lib.ptr(grad_dist1),
                                                _lib.ptr(idx1), _lib.ptr(grad_dist2), _lib.ptr(idx2), _lib.ptr(g1),
                                                _lib.ptr(g2), _lib.stream_handle())
    _lib.check(st, "nn_distance_grad")
    return g1, g2


class _NnDistanceFn(torch.autograd.Function):
    """Autograd glue equivalent to @ops.RegisterGradient('NnDistance') (tf_nndistance.py:35-41)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, i1, d2, i2 = nn_distance(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, i1, d2, i2

    @staticmethod
    def backward(ctx, gd1, _gi1, gd2, _gi2):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        g1, g2 = nn_distance_grad(xyz1, xyz2, gd1.contiguous(), i1, gd2.contiguous(), i2)
        return g1, g2


def nn_distance_autograd(xyz1, xyz2):
    """nn_distance with the registered gradient, for callers that differentiate through it."""
    return _NnDistanceFn.apply(xyz1, xyz2)


def microbench(which, iters=2000):
    """Calibration: ms for 2048x256 threads x 16*iters VALU instructions of kind `which`."""
    ms = C.c_float(0)
    st = _lib.lib().geoadv_microbench(int(which), int(iters), C.byref(ms), _lib.stream_handle())
    _lib.check(st, "microbench")
    return ms.value
