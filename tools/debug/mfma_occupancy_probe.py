"""How the encoder's matrix-instruction stream scales with what shares the SIMD and the chip (tools/probe/bf16x3.hip, tp_h2_kernel):
the f16x2 loop shape (12 MFMAs + 8 LDS fragment reads per step) with one / two workgroups per CU (one / two waves per SIMD), the three
products of an accumulator back to back (the kernel's order) or the four accumulators interleaved, on 1 ... 512 workgroups."""
import ctypes as C, json, os, sys
import torch
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "probe", "libgeoadv_probe_bf16x3.so")
lib = C.CDLL(so); lib.bf16x3_last_error.restype = C.c_char_p
lib.bf16x3_throughput.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
for fill, name, per_cu in ((300, "one workgroup per CU, products back to back", 1), (320, "one per CU, accumulators interleaved", 1),
                           (310, "two per CU, back to back", 2), (330, "two per CU, interleaved", 2), (313, "two per CU, back to back, 3 VALU per MFMA", 2),
                           (333, "two per CU, interleaved, 3 VALU per MFMA", 2)):
    for blocks in (1, 2, 64, 128, 256, 512, 1024):
        ms, g = C.c_float(0), C.c_float(0)
        rc = lib.bf16x3_throughput(fill, 0, blocks, 44, 2, 100, 0, C.byref(ms), C.byref(g), C.c_void_p(st))
        assert rc == 0, lib.bf16x3_last_error()
        mfmas_per_simd = 2 * 44 * 12 * max(1, -(-blocks // 256))            # waves of one SIMD of the busiest CU, one after / beside the other
        print(json.dumps({"form": name, "workgroups": blocks, "ms": round(ms.value, 5),
                          "pipe_cycles_per_mfma_at_2p4GHz": round(ms.value * 1e-3 * 2.4e9 / mfmas_per_simd, 1),
                          "f16_tflops": round(blocks * 4 * 2 * 44 * 12 * 32768 / ms.value / 1e9, 1)}), flush=True)
