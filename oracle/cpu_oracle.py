"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Loads oracle/_build/libgeoadv_oracle.so (our C restatement, geoadv_oracle.c) and, when it has
been built, oracle/_ref/libgeoadv_ref.so (the reference's own CPU functions compiled by
oracle/build_ref.sh).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; geometric_adv_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)


def _fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_F)


def _ip(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_I)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def build(quiet=True):
    """(Re)build the oracle libraries with oracle/Makefile."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(path):
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


class Oracle:
    """Our C restatement.  `omp=True` loads the OpenMP build (cpu_baseline 'all cores' leg)."""

    def __init__(self, omp=False):
        name = "libgeoadv_oracle_omp.so" if omp else "libgeoadv_oracle.so"
        self.lib = _load(os.path.join(_HERE, "_build", name))

    def nn_distance(self, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1 = np.empty((b, n), np.float32); i1 = np.empty((b, n), np.int32)
        d2 = np.empty((b, m), np.float32); i2 = np.empty((b, m), np.int32)
        self.lib.oracle_nn_distance(b, n, m, _fp(xyz1), _fp(xyz2), _fp(d1), _ip(i1), _fp(d2), _ip(i2))
        return d1, i1, d2, i2

    def nn_distance_grad(self, xyz1, xyz2, gd1, idx1, gd2, idx2):
        xyz1, xyz2, gd1, gd2 = _f32(xyz1), _f32(xyz2), _f32(gd1), _f32(gd2)
        idx1, idx2 = _i32(idx1), _i32(idx2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = np.empty_like(xyz1); g2 = np.empty_like(xyz2)
        self.lib.oracle_nn_distance_grad(b, n, m, _fp(xyz1), _fp(xyz2), _fp(gd1), _ip(idx1),
                                         _fp(gd2), _ip(idx2), _fp(g1), _fp(g2))
        return g1, g2

    def approx_match(self, xyz1, xyz2):
        """match in the reference CPU layout (b, n, m)."""
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = np.empty((b, n, m), np.float32)
        self.lib.oracle_approx_match(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match))
        return match

    def match_cost(self, xyz1, xyz2, match):
        xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = np.empty((b,), np.float32)
        self.lib.oracle_match_cost(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(cost))
        return cost

    def match_cost_grad(self, xyz1, xyz2, match):
        xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = np.empty_like(xyz1); g2 = np.empty_like(xyz2)
        self.lib.oracle_match_cost_grad(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(g1), _fp(g2))
        return g1, g2

    def selection_sort(self, k, dist):
        dist = _f32(dist)
        b, m, n = dist.shape
        idx = np.empty((b, m, n), np.int32); val = np.empty((b, m, n), np.float32)
        self.lib.oracle_selection_sort(b, n, m, int(k), _fp(dist), _ip(idx), _fp(val))
        return idx, val

    def query_ball_point(self, radius, nsample, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = np.zeros((b, m, nsample), np.int32); cnt = np.zeros((b, m), np.int32)
        self.lib.oracle_query_ball_point(b, n, m, C.c_float(radius), int(nsample), _fp(xyz1), _fp(xyz2),
                                         _ip(idx), _ip(cnt))
        return idx, cnt

    def group_point(self, points, idx):
        points, idx = _f32(points), _i32(idx)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = np.empty((b, m, ns, c), np.float32)
        self.lib.oracle_group_point(b, n, c, m, ns, _fp(points), _ip(idx), _fp(out))
        return out

    def group_point_grad(self, points, idx, grad_out):
        points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
        b, n, c = points.shape
        _, m, ns = idx.shape
        gp = np.empty((b, n, c), np.float32)
        self.lib.oracle_group_point_grad(b, n, c, m, ns, _fp(grad_out), _ip(idx), _fp(gp))
        return gp

    def knn_point(self, k, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        val = np.empty((b, m, k), np.float32); idx = np.empty((b, m, k), np.int32)
        self.lib.oracle_knn_point(b, n, m, int(k), _fp(xyz1), _fp(xyz2), _fp(val), _ip(idx))
        return val, idx

    def knn_dists(self, pc, k):
        pc = _f32(pc)
        b, n, _ = pc.shape
        out = np.empty((b, n, k), np.float32)
        self.lib.oracle_knn_dists(b, n, int(k), _fp(pc), _fp(out))
        return out


class Reference:
    """The reference's own CPU functions (oracle/_ref/libgeoadv_ref.so, C++-mangled symbols).
    Exists only where oracle/build_ref.sh has run (the build container; the .so travels to the
    GPU box).  Used to pin the oracle and as the 'reference' CPU baseline for the Chamfer op."""

    PATH = os.path.join(_HERE, "_ref", "libgeoadv_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)

    def nnsearch(self, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d = np.empty((b, n), np.float32); i = np.empty((b, n), np.int32)
        self.lib._Z8nnsearchiiiPKfS0_PfPi(b, n, m, _fp(xyz1), _fp(xyz2), _fp(d), _ip(i))
        return d, i

    def nn_distance(self, xyz1, xyz2):
        d1, i1 = self.nnsearch(xyz1, xyz2)
        d2, i2 = self.nnsearch(xyz2, xyz1)
        return d1, i1, d2, i2

    def nn_distance_grad(self, xyz1, xyz2, gd1, idx1, gd2, idx2):
        xyz1, xyz2, gd1, gd2 = _f32(xyz1), _f32(xyz2), _f32(gd1), _f32(gd2)
        idx1, idx2 = _i32(idx1), _i32(idx2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = np.empty_like(xyz1); g2 = np.empty_like(xyz2)
        self.lib._Z19nndistance_grad_cpuiiiPKfS0_S0_PKiS0_S2_PfS3_(
            b, n, m, _fp(xyz1), _fp(xyz2), _fp(gd1), _ip(idx1), _fp(gd2), _ip(idx2), _fp(g1), _fp(g2))
        return g1, g2

    def approx_match(self, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = np.empty((b, n, m), np.float32)
        self.lib._Z15approxmatch_cpuiiiPKfS0_Pf(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match))
        return match

    def match_cost(self, xyz1, xyz2, match):
        xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = np.empty((b,), np.float32)
        self.lib._Z13matchcost_cpuiiiPKfS0_S0_Pf(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(cost))
        return cost

    def match_cost_grad(self, xyz1, xyz2, match):
        xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1 = np.zeros_like(xyz1)        # zero-initialised: see geoadv_oracle.c on :108-109
        g2 = np.empty_like(xyz2)
        self.lib._Z17matchcostgrad_cpuiiiPKfS0_S0_PfS1_(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(g1), _fp(g2))
        return g1, g2

    def selection_sort(self, k, dist):
        """NB the reference CPU twin printf()s every row; stdout is redirected around the call."""
        dist = _f32(dist)
        b, m, n = dist.shape
        idx = np.zeros((b, m, n), np.int32); val = np.zeros((b, m, n), np.float32)
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            self.lib._Z18selection_sort_cpuiiiiPKfPiPf(b, n, m, int(k), _fp(dist), _ip(idx), _fp(val))
            C.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1); os.close(saved); os.close(devnull)
        return idx, val

    def query_ball_point(self, radius, nsample, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = np.zeros((b, m, nsample), np.int32)
        self.lib._Z20query_ball_point_cpuiiifiPKfS0_Pi(b, n, m, C.c_float(radius), int(nsample),
                                                      _fp(xyz1), _fp(xyz2), _ip(idx))
        return idx

    def group_point(self, points, idx):
        points, idx = _f32(points), _i32(idx)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = np.empty((b, m, ns, c), np.float32)
        self.lib._Z15group_point_cpuiiiiiPKfPKiPf(b, n, c, m, ns, _fp(points), _ip(idx), _fp(out))
        return out

    def group_point_grad(self, points, idx, grad_out):
        points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
        b, n, c = points.shape
        _, m, ns = idx.shape
        gp = np.zeros((b, n, c), np.float32)
        self.lib._Z20group_point_grad_cpuiiiiiPKfPKiPf(b, n, c, m, ns, _fp(grad_out), _ip(idx), _fp(gp))
        return gp
