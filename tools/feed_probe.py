"""What delivering the B operand costs an fp32 MFMA chain (tools/probe/feed.hip): buffer-load ring vs LDS copy vs none, one / two /
four row blocks per B fragment, one or two workgroups per CU.   python tools/feed_probe.py"""
import ctypes as C, json, os, subprocess
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
so = os.path.join(HERE, "libgeoadv_probe_feed.so")
src = os.path.join(HERE, "feed.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so, src], check=True)
lib = C.CDLL(so)
lib.geoadv_probe_feed.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
names = {0: "B by buffer loads", 1: "B from LDS", 2: "no B loads"}
for mode, rm, depth in [(2, 1, 4), (2, 2, 4), (2, 4, 4), (0, 1, 4), (0, 2, 4), (0, 4, 4), (0, 1, 8), (0, 2, 8), (1, 1, 4), (1, 2, 4), (1, 4, 4)]:
    for wgs in (1, 2):
        ms, tf = C.c_float(0), C.c_double(0)
        best = None
        for _ in range(3):
            rc = lib.geoadv_probe_feed(mode, rm, depth, wgs, 400 // rm, C.byref(ms), C.byref(tf))
            if rc != 0:
                break
            best = tf.value if best is None else max(best, tf.value)
        print(json.dumps({"B": names[mode], "row_blocks_per_fragment": rm, "ring_depth": depth, "workgroups_per_cu": wgs, "rc": rc,
                          "TFLOP_per_s": None if best is None else round(best, 1), "frac_of_157.3": None if best is None else round(best / 157.3, 3)}))

lib.geoadv_probe_feed_layers.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double)]
skews = {0: "none", 1: "odd workgroups half a chain late", 2: "second half of the grid late", 3: "bit 3 of the index late"}
for kg, nv, bar in [(16, 0, 0), (16, 0, 1), (16, 2, 0), (16, 2, 1), (16, 6, 1), (32, 2, 1), (32, 6, 1), (8, 2, 1), (8, 6, 1)]:
    for wgs, skew in ((1, 0), (2, 0), (2, 1), (2, 2), (2, 3)):
        ms, tf = C.c_float(0), C.c_double(0)
        best = None
        for _ in range(3):
            rc = lib.geoadv_probe_feed_layers(kg, nv, bar, wgs, 25600 // kg, skew, C.byref(ms), C.byref(tf))
            if rc != 0:
                break
            best = tf.value if best is None else max(best, tf.value)
        print(json.dumps({"probe": "layers", "k_groups_per_chain": kg, "valu_per_accumulator_register": nv + 1, "barriers": bool(bar), "workgroups_per_cu": wgs,
                          "skew": skews[skew], "rc": rc, "TFLOP_per_s": None if best is None else round(best, 1),
                          "frac_of_157.3": None if best is None else round(best / 157.3, 3)}))
