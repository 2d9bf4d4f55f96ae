"""Chamfer differences on the matrix pipe (tools/probe/mfma_sub.hip): the scan's per-pair arithmetic all-VALU against 3 MFMAs + VALU,
at 1 and 2 workgroups per CU, and the bit check of the MFMA's differences against v_sub.   python tools/mfma_sub_probe.py"""
import ctypes as C, json, os, subprocess
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
so = os.path.join(HERE, "libgeoadv_probe_mfmasub.so")
src = os.path.join(HERE, "mfma_sub.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-shared", "-o", so, src], check=True)
lib = C.CDLL(so)
lib.geoadv_probe_mfma_sub.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
steps = 4096
for blocks in (256, 512, 1024):
    row = {"workgroups": blocks, "steps": steps}
    for which, name in ((0, "valu"), (1, "mfma")):
        out = (C.c_double * 2)()
        rc = lib.geoadv_probe_mfma_sub(which, blocks, steps, 0, out)
        pairs = blocks * 8 * steps * 1024.0
        row[name] = {"rc": rc, "ms": round(out[0], 4), "Tpair_per_s": round(pairs / (out[0] * 1e-3) / 1e12, 3), "bit_mismatches_random": int(out[1])}
    row["mfma_over_valu"] = round(row["valu"]["ms"] / row["mfma"]["ms"], 3)
    print(json.dumps(row))
out = (C.c_double * 2)()
rc = lib.geoadv_probe_mfma_sub(1, 256, 16, 1, out)
print(json.dumps({"bit_check_special_values": {"rc": rc, "mismatches_of_%d" % (4096 * 64 * 16): int(out[1])}}))
