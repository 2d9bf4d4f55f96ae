"""Hand-off latency between two workgroups of one launch (tools/probe/handoff.hip): flag visibility and data read-back, on the
same XCD and across XCDs, with plain and with agent-scope (sc1) data accesses.   python tools/handoff_probe.py"""
import ctypes as C, json, os, subprocess
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
so = os.path.join(HERE, "libgeoadv_probe_handoff.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "handoff.hip")], check=True)
lib = C.CDLL(so)
lib.geoadv_probe_handoff.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for sc1 in (0, 1):
    for shift in (0, 1, 4):
        out = (C.c_double * 40)()
        rc = lib.geoadv_probe_handoff(sc1, shift, 64, out)
        rows = [out[5 * p:5 * p + 5] for p in range(8)]
        print(json.dumps({"data_accesses": "sc1 (agent scope)" if sc1 else "plain stores / volatile loads", "consumer_shift": shift, "rc": rc,
                          "xcc_pairs": [[int(r[0]), int(r[1])] for r in rows], "mismatches": int(sum(r[2] for r in rows)),
                          "flag_latency_us_median_per_pair": [round(r[3] / 100.0, 2) for r in rows],
                          "data_read_us_median_per_pair": [round(r[4] / 100.0, 2) for r in rows]}))
