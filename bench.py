#!/usr/bin/env python3
"""bench.py -- attack-iterations/sec of the geometric adversarial attack loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): B = 32 clouds of N = 2048 points per batch, output-space
attack (loss_adv_type = chamfer, loss_dist_type = chamfer, dist_weight 1.0, lr 0.01), synthetic
uniform clouds and seeded random-init weights of the reference architecture.  A "step" is one attack
iteration on one batch: Adam step on pert + the metrics of the updated pert (+ keep-best for the last
20 % of the run, like thresh 400 of 500).  Multi-GPU: every rank attacks its OWN batch of 32 (the
reference walks examples in independent batches, adv_ae.py:166-177) -- weak scaling, no data-path
collective; the final per-cloud loss scalars are all-gathered once (RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, N = 32, 2048
ENC_FLOP_PER_POINT = 2 * 90304            # 2 * (3*64 + 64*128 + 128*128 + 128*256 + 256*128)  
PEAK_MFMA_F32_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
PMC_HBM_FILE = os.path.join(ROOT, "profiles", "r01_v15_pmc_hbm.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
PMC_SQ_FILE = os.path.join(ROOT, "profiles", "r01_v15_pmc_sq.json")       # rocprofv3 --pmc SQ_* pass


def _encoder_entry(d):
    """The attack loop's forward kernel in a tools/pmc_summary.py file: the mask-writing instantiation
    encoder_fwd2_kernel<false, true> (the <false, false> one is the plain geoadv_ae_forward)."""
    names = [n for n in d if "encoder_fwd" in n]
    names.sort(key=lambda n: ("true>" not in n, n))
    return d[names[0]]


def chamfer_valu(avg_ms):
    """VALU issue rate of the all-pairs Chamfer kernels of one step: wave-level VALU instructions from the committed PMC
    pass (SQ_INSTS_VALU, profiles/r01_v9_pmc_sq.json) x 64 lanes / the measured time, beside two yardsticks measured on the
    box: geoadv_microbench (profiles/r01_probe_valu_chamfer_v1.json: 52.1 T lane-instr/s for an alternating v_mul/v_add
    stream, 32 T for a single instruction type) and the best rate a real kernel sustains (the two-scan Chamfer at B=256: 63 T)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_v9_pmc_sq.json")))
        insts = d["geoadv::chamfer_sym_kernel"]["SQ_INSTS_VALU"]["mean"] + d["geoadv::chamfer_sym_finish_kernel"]["SQ_INSTS_VALU"]["mean"]
        grid = d["geoadv::latent_decode_and_grid_kernel"]["SQ_INSTS_VALU"]["mean"] - d["geoadv::latent_decode_kernel"]["SQ_INSTS_VALU"]["mean"]
        ceil = json.load(open(os.path.join(ROOT, "profiles", "r01_probe_valu_chamfer_v1.json")))["valu_mul+add"]["Tinstr_lane_per_s"]
        rate = insts * 64.0 / (avg_ms * 1e-3) / 1e12
        return {"T_lane_instr_per_s": rate, "microbench_mul_add_T_lane_instr_per_s": ceil,
                "best_sustained_by_a_kernel_T_lane_instr_per_s": 63.0,      # two-scan Chamfer at B=256 (7.0 T pair-evals/s x 9)
                "frac_of_best_sustained": rate / 63.0, "wave_instr_per_step_pmc": insts,
                "paired_grid_search_wave_instr_per_step_pmc": grid}
    except Exception:               # pragma: no cover
        return None


def pmc_traffic_bytes():
    """HBM-side bytes per encoder launch from the committed PMC passes (separate --pmc runs, guide recipe):
    FETCH_SIZE is in KiB and under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM),
    WRITE_SIZE is exact.  Returns (bytes, note) or (None, reason)."""
    try:
        k = _encoder_entry(json.load(open(PMC_HBM_FILE)))
        fetch, write = k["FETCH_SIZE"]["mean"], k["WRITE_SIZE"]["mean"]
        return (2.0 * fetch + write) * 1024.0, "profiles/r01_v15_pmc_hbm.json: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch"
    except Exception as e:          # pragma: no cover
        return None, "no PMC profile: %s" % e


def pmc_mfma_util():
    try:
        k = _encoder_entry(json.load(open(PMC_SQ_FILE)))
        return k["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / (k["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
    except Exception:               # pragma: no cover
        return None


def clouds(seed, b, n):
    rng = np.random.default_rng(seed)
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def cpu_baseline(weights, x, gt, iters=5, gpu_clouds=None):
    """The oracle's attack iteration (numpy fp32 GEMMs on all cores + the single-threaded C Chamfer
    restatement, i.e. the reference's threading: NnDistanceOp::Compute is single-threaded,
    tf_nndistance.cpp:79-80) in the REFERENCE schedule: step (fwd+bwd+Adam) + a second metrics
    forward (adv_ae.py:217-221).  Bounded sample: `iters` iterations after one warm-up."""
    from geometric_adv_amd import weights as W
    from oracle.attack_model import AEModel, AttackModel
    from geometric_adv_amd.adversary import init_pert_value
    model = AEModel(W.canonical(weights, N), N, np.float32)
    am = AttackModel(model, x, gt, None, np.ones(B, np.float32))
    am.init_pert(init_pert_value(B, N))
    am.step(); am.forward()
    t0 = time.perf_counter()
    for _ in range(iters):
        am.step()
        am.forward()
    dt = time.perf_counter() - t0
    # second leg (SURVEY 8d ii, shown for honesty): the same iterations with the Chamfer restatement
    # parallelised over the clouds of the batch with OpenMP -- something the reference op does not do
    import oracle.attack_model as am_mod
    from oracle.cpu_oracle import Oracle
    all_cores = None
    try:
        saved, am_mod._oracle = am_mod._oracle, Oracle(omp=True)
        am.step(); am.forward()
        t1 = time.perf_counter()
        for _ in range(iters):
            am.step()
            am.forward()
        dt2 = time.perf_counter() - t1
        am_mod._oracle = saved
        all_cores = {"value": iters / dt2, "sec_per_iteration": dt2 / iters,
                     "note": "Chamfer restatement with OpenMP over the clouds of the batch (not what the reference does)"}
    except OSError:                 # OpenMP build of the oracle missing
        pass
    parity = None
    if gpu_clouds is not None:      # the oracle as the checker of the metric's second half: Chamfer rel-err and exact indices
        from geometric_adv_amd import ops
        import torch
        p, q = gpu_clouds           # GPU tensors (recon, target) of the attacked batch, first clouds only
        d1, i1, d2, i2 = [t.cpu().numpy() for t in ops.nn_distance(p, q)]
        o1, oi1, o2, oi2 = Oracle().nn_distance(p.cpu().numpy(), q.cpu().numpy())
        loss_gpu = d1.mean(axis=1, dtype=np.float64) + d2.mean(axis=1, dtype=np.float64)
        loss_ref = o1.mean(axis=1, dtype=np.float64) + o2.mean(axis=1, dtype=np.float64)
        parity = {"clouds": int(p.shape[0]), "chamfer_loss_rel_err_max": float(np.abs(loss_gpu / loss_ref - 1.0).max()),
                  "dist_bit_exact": bool(np.array_equal(d1, o1) and np.array_equal(d2, o2)),
                  "idx_exact": bool(np.array_equal(i1, oi1) and np.array_equal(i2, oi2))}
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": iters / dt, "unit": "attack-iterations/sec", "cores": os.cpu_count(), "kind": "port", "parity": parity,
            "cpu_model": cpu_model, "oracle_flags": "gcc -O2 -ffp-contract=off (oracle/Makefile); numpy %s BLAS" % np.__version__,
            "sample": "%d iterations of config 2 (B=32, N=2048) after 1 warm-up, reference schedule (2 forwards/iter); "
                      "numpy fp32 GEMMs on all cores, Chamfer single-threaded C (gcc -O2 -ffp-contract=off)" % iters,
            "sec_per_iteration": dt / iters, "all_cores": all_cores}


def training_leg(dev, steps=30, batch=50):
    """AE training steps/s at default_train_params (batch 50 x 2048 points, lr 0.0005), synthetic clouds."""
    import torch
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    tr = PointNetAETrainer(initial_weights(N, seed=1), N, batch_size=batch, device=dev)
    xb = torch.as_tensor(clouds(77, batch, N)).to(dev)
    for _ in range(3):
        tr.partial_fit(xb, want_recon=False, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.partial_fit(xb, want_recon=False, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop = 3 * 2.0 * batch * N * 90304 + 3 * 2.0 * batch * (98304 + 768 * N)
    return {"steps_per_sec": 1.0 / dt, "ms_per_step": dt * 1e3, "batch": batch, "n_points": N,
            "frac_of_fp32_mfma_peak_end_to_end": flop / dt / (PEAK_MFMA_F32_TFLOPS * 1e12)}


def slots_leg(dev, weights, ae, slots=2, iters=300):
    """Configuration.batch_slots: `slots` independent B = 32 batches attacked concurrently on this GPU (own handle, stream
    and host thread each, AdvAE._attack_slots): aggregate attack iterations/s.  Reported beside the headline, which
    stays one batch at a time."""
    import threading
    import torch
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    ats, streams = [], []
    for s in range(slots):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            at = AdvAE("adversary", Configuration(batch_size=B, n_points=N, weights=weights, num_iterations=iters + 20,
                                                  num_iterations_thresh=10 ** 6), device=dev, ae=ae)
            x, gt = clouds(500 + 2 * s, B, N), clouds(501 + 2 * s, B, N)
            at.set_inputs(x, gt, None, 1.0)
            at.init_pert(None, reset_optimizer=True)
            at.run(0, 20, 10 ** 6)
        ats.append(at)
        streams.append(st)
    torch.cuda.synchronize()

    def work(at, st):
        with torch.cuda.stream(st):
            at.run(20, iters, 10 ** 6)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(a, s)) for a, s in zip(ats, streams)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"slots": slots, "batch": B, "attack_iterations_per_sec_all_slots": slots * iters / dt,
            "ms_per_iteration_per_slot": dt / iters * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--slots", type=int, default=0,
                    help="also measure S concurrent batch slots (secondary.batch_slots; off by default: its overlapping "
                         "launches would distort the per-kernel averages of a rocprofv3 run of this command)")
    args = ap.parse_args()

    import torch
    from geometric_adv_amd import _lib, dist as gdist
    rank, world, local = gdist.env_rank()
    if not os.path.exists(_lib.LIB_PATH):                     # clean checkout: build in-tree first (no fallback path exists)
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:                                                 # the other ranks wait for rank 0's build
            for _ in range(600):
                if os.path.exists(_lib.LIB_PATH):
                    break
                time.sleep(0.5)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with --nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = "nccl"
    try:
        gdist.init("nccl")                       # RCCL over xGMI
        if world > 1:
            gdist.barrier()
    except Exception as e:                       # keep the scaling run alive if RCCL cannot come up
        sys.stderr.write("bench.py: RCCL init failed (%s); falling back to gloo for the KB-sized gather\n" % e)
        backend = "gloo"
        import torch.distributed as tdist
        if tdist.is_initialized():
            tdist.destroy_process_group()
        gdist.init("gloo")

    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE

    K, Wm = args.steps, args.warmup
    weights = W.synthetic_weights(N, seed=7)
    x = clouds(1000 + 2 + 17 * rank, B, N)            # source batch of this rank
    gt = clouds(2000 + 2 + 17 * rank, B, N)           # target batch of this rank
    total = Wm + K
    thresh = Wm + int(0.8 * K) + 1
    conf = Configuration(batch_size=B, n_points=N, weights=weights, loss_adv_type="chamfer", loss_dist_type="chamfer",
                         dist_weight_list=[1.0], num_iterations=total, num_iterations_thresh=thresh, learning_rate=0.01)
    ae = PointNetAE(weights, N, device=dev)
    ref = torch.as_tensor(ae.get_loss_per_pc(gt)).to(dev)      # target_ae_loss_ref
    tz = ae.transform(gt)
    at = AdvAE("adversary", conf, device=dev, ae=ae)
    at.set_inputs(x, gt, tz, 1.0)
    at.init_pert(None, reset_optimizer=True)

    at.run(0, Wm, thresh)                                      # W untimed warm-up steps
    at.profile(["encoder_fwd"], stride=4)                      # HIP events on the launch stream around every 4th launch of the
                                                               # dominant kernel in the timed region (each pair costs ~1 % if on all)
    torch.cuda.synchronize()
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    at.run(Wm, K, thresh)                                      # exactly K timed steps, no host sync inside
    metrics, _, _ = at.get_best(ref)
    gathered = gdist.all_gather_examples(metrics[None] if backend == "nccl" else metrics[None].cpu(), axis=1)   # final loss scalars only
    torch.cuda.synchronize()
    gdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = gdist.max_over_ranks(dt, device=dev if backend == "nccl" else "cpu")
    prof = at.profile_read()
    at.profile(False)

    if rank != 0:
        return
    # per-kernel breakdown in a separate, untimed pass
    at.profile(True)
    at.run(total, 50, total + 1000)
    torch.cuda.synchronize()
    breakdown = {k: (ms / max(n_, 1)) for k, (n_, ms) in at.profile_read().items()}
    at.profile(False)

    enc_n, enc_ms = prof["encoder_fwd"]
    enc_avg_ms = enc_ms / max(enc_n, 1)
    enc_flop = ENC_FLOP_PER_POINT * B * N                       # algorithmic flop per launch
    enc_tflops = enc_flop / (enc_avg_ms * 1e-3) / 1e12
    traffic, traffic_note = pmc_traffic_bytes()
    ch_n, ch_avg_ms = 50, breakdown["chamfer_fwd"]              # (from the untimed per-class pass: every pair of events costs ~1 %)
    pruned = os.environ.get("GEOADV_CHAMFER_PRUNE", "1") != "0"
    # all-pairs kernels: nn_distance(recon, target) always; nn_distance(adv, x) too unless the paired grid search has it
    ch_pairs = (2.0 if pruned else 4.0) * B * N * N             # problems x 2 directions per step
    ch_bytes = (1 if pruned else 2) * 20.0 * B * (N + N)        # 20*B*(N+M) per nn_distance call
    out = {
        "metric": "attack-iterations/sec (B=32, N=2048) at 1/2/4/8 GPUs; Chamfer rel-err vs ref",
        "value": world * K / dt, "unit": "attack-iterations/sec", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: B=32 random clouds x N=2048, output-space attack (chamfer/chamfer, "
                               "dist_weight 1.0, lr 0.01), one batch per GPU", "batch_per_gpu": B, "n_points": N,
                   "global_batch": B * world, "parallelism": "batches sharded, dp%d, no data-path collective" % world,
                   "collective_backend": backend if world > 1 else "none",
                   "thresh_fraction": 0.8},
        "roofline": {"bound": "mfma", "kernel": "encoder_fwd2_kernel", "achieved": enc_tflops, "peak": PEAK_MFMA_F32_TFLOPS,
                     "unit": "TFLOP/s", "frac": enc_tflops / PEAK_MFMA_F32_TFLOPS, "traffic": traffic, "traffic_source": traffic_note,
                     "mfma_pipe_utilisation_pmc": pmc_mfma_util(),
                     "avg_launch_ms": enc_avg_ms, "launches_timed": enc_n, "algorithmic_flop_per_launch": enc_flop},
        "roofline_chamfer": {"bound": "valu", "kernel": "chamfer_sym_kernel + chamfer_sym_finish_kernel: nn_distance(recon, target), both "
                                                         "directions from one distance evaluation per pair" +
                                                         ("; nn_distance(adv, x) is answered exactly by the paired grid search inside the "
                                                          "latent_decode launch (decoder_fwd class)" if pruned else " (and nn_distance(adv, x))"),
                             "avg_launch_ms": ch_avg_ms,
                             "launches_timed": ch_n, "achieved_Tpair_per_s": ch_pairs / (ch_avg_ms * 1e-3) / 1e12,
                             "valu": chamfer_valu(ch_avg_ms),
                             "algorithmic_bytes_per_launch": ch_bytes,
                             "achieved_hbm_GBps": ch_bytes / (ch_avg_ms * 1e-3) / 1e9, "hbm_peak_GBps": PEAK_HBM_GBS},
        "kernel_ms_per_iteration": breakdown,
        "final_mean_target_recon_error": float(gathered[0, :, 4].mean().item()),
    }
    if world == 1:                  # the widened row f-4, measured beside the headline (not part of `value`)
        out["secondary"] = {"ae_training_step": training_leg(dev)}
        if args.slots > 1:
            out["secondary"]["batch_slots"] = slots_leg(dev, weights, ae, args.slots)
    if world == 1 and not args.no_cpu_baseline:
        _, adv_best, recon_best = at.get_best(ref)
        out["cpu_baseline"] = cpu_baseline(weights, x, gt, args.cpu_iters,
                                           gpu_clouds=(recon_best[:4].contiguous(), torch.as_tensor(gt[:4]).to(dev)))
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
