"""The reference-weight mode of approx-EMD evaluates expf the way glibc does (csrc/emd.hip, PairWeight<true>::w).  Its
constants are checked here without a GPU: the table in the source against 2^(i/32), and the algorithm -- restated in numpy
float64 from the constants parsed out of the .hip file -- bit for bit against the host libm over the argument range the
CPU op produces (tf_approxmatch.cpp:46: level * d2 <= 0), including the denormal results and the underflow cut."""
import ctypes
import os
import re
import struct
from decimal import Decimal, getcontext

import numpy as np

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geometric_adv_amd", "csrc", "emd.hip")


def _source_constants():
    s = open(SRC).read()
    body = s[s.index("EXPF_TAB[32] = {"):]
    body = body[:body.index("};")]
    tab = [int(h, 16) for h in re.findall(r"0x([0-9a-f]{16})ull", body)]
    w = s[s.index("template <> struct PairWeight<true>"):]
    w = w[:w.index("};\n")]
    hexf = {k: float.fromhex(v) for k, v in re.findall(r"(INV_LN2_N|SHIFT|C0|C1|C2) = (-?0x[0-9a-f.]+p[+-]?\d+)", w)}
    cut = float.fromhex(re.search(r"a < (-0x[0-9a-f.]+p\d+)f", w).group(1))
    return np.array(tab, dtype=np.uint64), hexf, cut


def test_table_is_two_to_the_i_over_32():
    tab, _, _ = _source_constants()
    assert len(tab) == 32
    getcontext().prec = 60
    for i in range(32):
        v = float(Decimal(2) ** (Decimal(i) / Decimal(32)))              # correctly rounded
        bits = struct.unpack("<Q", struct.pack("<d", v))[0]
        assert int(tab[i]) == (bits - (i << 47)) & 0xFFFFFFFFFFFFFFFF, i


def test_algorithm_equals_host_expf_bit_for_bit():
    tab, k, cut = _source_constants()
    inv, shift = k["INV_LN2_N"] * 32, k["SHIFT"]
    c0, c1, c2 = k["C0"] / 32 / 32 / 32, k["C1"] / 32 / 32, k["C2"] / 32
    rng = np.random.default_rng(0)
    x = np.concatenate([-(10 ** rng.uniform(-6, 2.02, 200000)), -rng.uniform(80, 110, 20000), [0.0, -0.0, -2e5]]).astype(np.float32)
    z = inv * x.astype(np.float64)
    kd = z + shift
    ki = kd.view(np.uint64)
    r = z - (kd - shift)
    with np.errstate(over="ignore", invalid="ignore"):
        s = (tab[(ki % np.uint64(32)).astype(np.int64)] + (ki << np.uint64(47))).view(np.float64)
        y = ((c0 * r + c1) * (r * r) + (c2 * r + 1.0)) * s
        got = np.where(x < np.float32(cut), np.float32(0), y.astype(np.float32))
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    want = np.array([libm.expf(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
