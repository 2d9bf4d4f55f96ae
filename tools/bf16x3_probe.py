"""Probe: the encoder's fp32 products emulated on the bf16 matrix pipe (tools/probe/bf16x3.hip) -- accuracy and throughput.

    python tools/bf16x3_probe.py > gpurun_out/bf16x3_probe.jsonl

accuracy:   |C - exact| / (|A| @ |B|) per element (exact = float64 on the host) of the fp32 MFMA chain, the 6-term and 9-term
            3 x bf16 forms (round-to-nearest and truncating splits) and the 3-term 2 x bf16 form, K = 64 / 128 / 256,
            A = relu(N(0, 1)) (activations), B = N(0, 0.1) (weights).
throughput: a wave-private encoder's loop shape, 512 workgroups of 4 waves x 2 units x 44 sixteen-k steps (the B = 32 headline
            launch: 65536 points = 2048 units of 32 points, 1056 MFMAs each), alone and with a VALU kernel of ~120 us between
            launches (the attack's duty cycle).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "probe", "libgeoadv_probe_bf16x3.so")
src = os.path.join(HERE, "probe", "bf16x3.hip")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-fno-slp-vectorize", "-shared", "-o", so, src], check=True)
lib = C.CDLL(so)
lib.bf16x3_last_error.restype = C.c_char_p
lib.bf16x3_accuracy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.bf16x3_throughput.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]


def check(rc):
    if rc:
        raise RuntimeError(lib.bf16x3_last_error().decode())


def accuracy():
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    names = {0: "f32_mfma_chain", 1: "bf16x3_6term_rne", 2: "bf16x3_9term_rne", 3: "bf16x3_6term_trunc", 4: "bf16x2_3term_rne", 5: "f16x2_3term", 6: "f16x2_4term", 7: "f16x2_3term_subnormal_second_pieces"}
    for K in (64, 128, 256):
        tiles = 512
        A = torch.randn(tiles, 32, K, generator=g).clamp_min(0).contiguous()
        B = (0.1 * torch.randn(K, 32, generator=g)).contiguous()
        exact = A.double() @ B.double()
        scale = A.double().abs() @ B.double().abs()
        Ad, Bd = A.to(dev), B.to(dev)
        row = {"probe": "accuracy", "K": K, "tiles": tiles, "unit": "|C - exact| / (|A| @ |B|), in units of 2^-24"}
        res = {}
        for mode, name in names.items():
            Cd = torch.zeros(tiles, 32, 32, device=dev)
            check(lib.bf16x3_accuracy(mode, Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), tiles, K, None))
            torch.cuda.synchronize()
            c = Cd.cpu().double()
            err = ((c - exact).abs() / scale) * 2.0 ** 24
            res[name] = c
            row[name] = {"max": round(float(err.max()), 4), "mean": round(float(err.mean()), 5), "rms": round(float((err ** 2).mean().sqrt()), 5)}
        # the straightforward fp32 chain in ascending k on the host (numpy float32 fma-free: products rounded) for scale
        c32 = torch.zeros(tiles, 32, 32)
        for k in range(K):
            c32 += A[:, :, k:k + 1] * B[k:k + 1, :].unsqueeze(0)
        err = ((c32.double() - exact).abs() / scale) * 2.0 ** 24
        row["host_f32_mul_add_chain"] = {"max": round(float(err.max()), 4), "mean": round(float(err.mean()), 5), "rms": round(float((err ** 2).mean().sqrt()), 5)}
        row["bits_equal_6term_rne_vs_f32_chain"] = float((res["bf16x3_6term_rne"] == res["f32_mfma_chain"]).double().mean())
        print(json.dumps(row), flush=True)


def throughput():
    st = torch.cuda.current_stream().cuda_stream
    for (fill, src, gap) in ((0, 1, 0), (3, 1, 0), (0, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0), (6, 0, 0), (3, 0, 40000), (0, 0, 40000), (3, 0, 80000)):
        ms, gms = C.c_float(0), C.c_float(0)
        check(lib.bf16x3_throughput(fill, src, 512, 44, 2, 200, gap, C.byref(ms), C.byref(gms), C.c_void_p(st)))
        mfmas = 512 * 4 * 2 * 44 * 24
        flops = mfmas * 32 * 32 * 16 * 2
        print(json.dumps({"probe": "throughput", "valu_per_mfma": fill, "weights_from": "lds" if src == 0 else "registers",
                          "gap_kernel_ms": round(gms.value, 4), "ms": round(ms.value, 5), "bf16_tflops": round(flops / ms.value / 1e9, 1),
                          "cycles_per_mfma_at_2p4GHz": round(ms.value * 1e-3 * 2.4e9 / (2 * 44 * 24) / 2, 2),
                          "emulated_f32_tflops": round(flops / 6 / ms.value / 1e9, 1),
                          "note": "512 workgroups x 4 waves x 2 units x 44 steps x 24 MFMAs = the B = 32 encoder's MFMA count (2 rounds on 256 CUs)"}), flush=True)


def chains():
    """Dependent against independent consecutive MFMAs, on a quarter of the chip (no clock give-back) and on all of it."""
    st = torch.cuda.current_stream().cuda_stream
    for blocks in (64, 256, 512):
        for fill, name in ((0, "four accumulators interleaved"), (100, "six products of an accumulator back to back"), (101, "one accumulator"),
                           (102, "six back to back + 3 VALU per MFMA")):
            ms, gms = C.c_float(0), C.c_float(0)
            check(lib.bf16x3_throughput(fill, 0, blocks, 44, 2, 100, 0, C.byref(ms), C.byref(gms), C.c_void_p(st)))
            rounds = max(1, blocks // 256)
            print(json.dumps({"probe": "chains", "order": name, "workgroups": blocks, "ms": round(ms.value, 5),
                              "cycles_per_mfma_at_2p4GHz": round(ms.value * 1e-3 * 2.4e9 / (2 * 44 * 24 * rounds), 2)}), flush=True)


def shapes():
    """32x32x16 against 16x16x32 for the same work (MFMA count doubles, each half the size), whole chip."""
    st = torch.cuda.current_stream().cuda_stream
    for blocks in (256, 512):
        for fill, name in ((0, "32x32x16"), (3, "32x32x16 + 3 VALU per MFMA"), (200, "16x16x32"), (201, "16x16x32 + 1.5 VALU per MFMA")):
            ms, gms = C.c_float(0), C.c_float(0)
            check(lib.bf16x3_throughput(fill, 0, blocks, 44, 2, 100, 0, C.byref(ms), C.byref(gms), C.c_void_p(st)))
            flops = blocks * 4 * 2 * 44 * 24 * 32 * 32 * 16 * 2
            print(json.dumps({"probe": "shapes", "shape": name, "workgroups": blocks, "ms": round(ms.value, 5), "bf16_tflops": round(flops / ms.value / 1e9, 1)}), flush=True)


def throughput_h2():
    """The f16x2 encoder's loop shape (tp_h2_kernel): 12 MFMAs and 8 fragment reads per step; random fp16 operands."""
    st = torch.cuda.current_stream().cuda_stream
    for (fill, gap) in ((300, 0), (303, 0), (306, 0), (303, 40000)):
        ms, gms = C.c_float(0), C.c_float(0)
        check(lib.bf16x3_throughput(fill, 0, 512, 44, 2, 200, gap, C.byref(ms), C.byref(gms), C.c_void_p(st)))
        mfmas = 512 * 4 * 2 * 44 * 12
        flops = mfmas * 32 * 32 * 16 * 2
        print(json.dumps({"probe": "throughput_f16x2", "valu_per_mfma": fill - 300, "weights_from": "lds", "gap_kernel_ms": round(gms.value, 4),
                          "ms": round(ms.value, 5), "f16_tflops": round(flops / ms.value / 1e9, 1),
                          "cycles_per_mfma_at_2p4GHz": round(ms.value * 1e-3 * 2.4e9 / (2 * 44 * 12) / 2, 2),
                          "emulated_f32_tflops": round(flops / 3 / ms.value / 1e9, 1),
                          "note": "512 workgroups x 4 waves x 2 units x 44 steps x 12 MFMAs = the B = 32 encoder's MFMA count in the f16x2 arithmetic"}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "accuracy"):
        accuracy()
    if what in ("all", "throughput"):
        throughput()
    if what in ("all", "chains"):
        chains()
    if what in ("all", "shapes"):
        shapes()
    if what in ("all", "f16x2"):
        throughput_h2()
