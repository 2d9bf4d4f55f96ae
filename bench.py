#!/usr/bin/env python3
"""bench.py -- attack-iterations/sec of the geometric adversarial attack loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 from a bare shell: this process starts N ranks itself (a child `python -m torch.distributed.run`, spawned before
anything here touches the GPU), relays rank 0's JSON line and exits with the children's status.  Under an external
torchrun (WORLD_SIZE set) it is one of the ranks.

Workload (BASELINE.json configs[1]): B = 32 clouds of N = 2048 points per batch, output-space attack (loss_adv_type =
chamfer, loss_dist_type = chamfer, dist_weight 1.0, lr 0.01), synthetic uniform clouds and seeded random-init weights
of the reference architecture.  A "step" is one attack iteration on one batch: Adam step on pert + the metrics of the
updated pert (+ keep-best for the last 20 % of a window, like thresh 400 of 500).

Timing: a WINDOW is exactly K steps between barrier + synchronize pairs (max over ranks); WINDOWS windows run back to
back and the MEDIAN window is reported (a single 20-step window is 4 ms -- shorter than the clock ramp of an idle chip).

Multi-GPU: `value` = weak scaling, every rank attacks its OWN batch of 32 (the reference walks examples in independent
batches, adv_ae.py:166-177; no data-path collective, the final per-cloud loss scalars are all-gathered once per window,
RCCL).  `strong_scaling` = ONE global batch of 32 split 32/N per rank (SURVEY 8e's definition), measured the same way.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, N = 32, 2048
ENC_FLOP_PER_POINT = 2 * 90304            # 2 * (3*64 + 64*128 + 128*128 + 128*256 + 256*128)  
PEAK_MFMA_F32_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_MFMA_BF16_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 32 cycles / SIMD at 2.4 GHz)
X3_PRODUCTS = 3                           # fp16 piece products per fp32 product of the encoder's default arithmetic, f16x2 (csrc/encoder_x3.h;
                                          # the dense fp16 and bf16 MFMA peaks are the same figure); bf16x3 takes 6
PEAK_ENCODER_X3_TFLOPS = PEAK_MFMA_BF16_TFLOPS / X3_PRODUCTS     # algorithmic fp32 TFLOP/s the fp16 pipe can deliver in that form
PEAK_HBM_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_encoder.json")     # tools/pmc_summary.py output + source hashes
PMC_CHAMFER_FILE = os.path.join(ROOT, "profiles", "r06_pmc_chamfer_hbm.json")   # the same for the Chamfer kernels of the plain B = 32 loop
CSRC = os.path.join(ROOT, "geometric_adv_amd", "csrc")
SWEEP_BATCHES = (16, 8, 4)                 # what a rank of the strong-scaling leg runs at 2 / 4 / 8 GPUs: timed in THIS run (small_batch_sweep)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=7, help="timed windows of --steps steps each; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--window-timeout", type=float, default=120.0,
                    help="wall-clock limit in seconds for the process-group start-up and for every timed window; a rank that "
                         "exceeds it prints what it was waiting for and exits 3 (no hang: a dead peer or a stuck collective "
                         "ends the run)")
    ap.add_argument("--run-timeout", type=float, default=1500.0,
                    help="self-launched N > 1 runs: wall-clock limit in seconds for all ranks together")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="(internal) child mode: bring up a ONE-rank RCCL group on cuda:0, run the collectives of the N > 1 path "
                         "on device tensors, print one JSON line")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip every secondary leg (other BASELINE configs, trained victim, training step, EMD, surfaces) and the "
                         "B = 16 / 8 / 4 windows of strong_scaling.measured_ms: the headline, its roofline and the CPU baseline only -- "
                         "what the counter passes of tools/collect_pmc.sh run")
    ap.add_argument("--only-leg", choices=["config2", "config3", "config4", "trained_victim", "training", "emd", "surfaces"],
                    help="run ONE secondary leg alone on cuda:0 and print its JSON (what `rocprofv3 --kernel-trace --stats -- python3 "
                         "bench.py --only-leg X` profiles: tools/collect_legs.sh -> profiles/r04_leg_*_kernel_stats.csv)")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="do not start the one-rank RCCL child at N = 1")
    ap.add_argument("--slots", type=int, default=-1,
                    help="also measure S concurrent batch slots (secondary.batch_slots).  Default: 2 and 3 slots, unless the run is "
                         "being profiled (a ROCPROF* / ROCP_* variable in the environment: overlapping launches would distort the "
                         "per-kernel averages of a rocprofv3 run of this command); 0: never")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# self-launch (parent side; must not touch the GPU)
# ------------------------------------------------------------------------------------------------------------------
def ensure_built():
    """Build libgeoadv.so BEFORE any rank starts (make only: nothing here loads the library or initialises HIP), so no
    rank can dlopen a half-linked file."""
    lib = os.path.join(ROOT, "geometric_adv_amd", "lib", "libgeoadv.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 4))], check=True, stdout=sys.stderr)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=sys.stderr)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    ensure_built()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["GEOADV_BENCH_CHILD"] = "1"
    if env.get("GEOADV_BENCH_SHARE_GPU") != "1":
        import torch                                   # device_count() does not initialise the GPU on this image
        ndev = torch.cuda.device_count()
        if args.gpus > ndev:
            sys.stderr.write("bench.py: --gpus %d but this node has %d GPU(s); one rank per GPU (RCCL refuses two ranks on one "
                             "device).  GEOADV_BENCH_SHARE_GPU=1 runs the ranks on cuda:0 with a gloo group instead.\n" % (args.gpus, ndev))
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)   # own process group: killable as one
    try:
        stdout, _ = p.communicate(timeout=args.run_timeout)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(p.pid, signal.SIGKILL)               # exactly the group started above
        p.wait()
        sys.stderr.write("bench.py: the %d-rank run exceeded --run-timeout %.0f s and was killed\n" % (args.gpus, args.run_timeout))
        sys.exit(124)
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in stdout.splitlines():
        if not ln.startswith('{"metric"'):
            sys.stderr.write(ln + "\n")
    if p.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (exit %d)\n" % (args.gpus, p.returncode))
        sys.exit(p.returncode or 1)
    print(lines[-1])
    sys.exit(0)


class Watchdog:
    """Wall-clock limit for the phases of a rank that wait on OTHER ranks (process-group start-up, every timed window): when a
    phase overruns, the rank says what it was doing and exits 3 -- torchrun then ends the other ranks -- instead of sitting in a
    collective for ever.  A timer thread and os._exit: nothing is exec'ed, no new process is started from a GPU process."""

    def __init__(self, rank, limit_s):
        import threading
        self.rank, self.limit, self.deadline, self.label = rank, float(limit_s), None, ""
        self.lock = threading.Lock()
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def arm(self, label, scale=1.0):
        with self.lock:
            self.label, self.deadline = label, time.monotonic() + self.limit * scale

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _watch(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                label = self.label
            if late:
                sys.stderr.write("bench.py: rank %d exceeded --window-timeout (%.0f s) in: %s -- exiting 3\n" % (self.rank, self.limit, label))
                sys.stderr.flush()
                os._exit(3)


# ------------------------------------------------------------------------------------------------------------------
# helpers (rank side)
# ------------------------------------------------------------------------------------------------------------------
def clouds(seed, b, n):
    import numpy as np
    rng = np.random.default_rng(seed)
    return (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)


def source_hashes(files):
    out = {}
    for f in files:
        try:
            out[f] = hashlib.sha1(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:12]
        except OSError:
            out[f] = None
    return out


def pmc_encoder():
    """HBM-side bytes per encoder launch and MFMA pipe utilisation from the committed PMC passes (separate --pmc runs;
    FETCH_SIZE is in KiB and under-reports wide coalesced reads by 2x on gfx950, WRITE_SIZE is exact: MI355X_MICROARCH.md).
    The file records the sha1 of the kernel sources it was taken at; if they differ from the tree, the counters are stale
    and nothing derived from them is reported."""
    try:
        d = json.load(open(PMC_FILE))
    except Exception as e:
        return None, None, "no PMC profile (%s)" % e
    want = d.get("_source_sha1", {})
    if not want or source_hashes(sorted(want)) != want:
        return None, None, "%s was taken at different kernel sources: dropped" % os.path.basename(PMC_FILE)
    names = [n for n in d if "encoder_fwd" in n]
    names.sort(key=lambda n: ("encoder_fwd3_kernel<2, true>" not in n, "encoder_fwd3_kernel<3, true>" not in n, "<true, 64>" not in n, n))      # the loop's instantiation at B = 32: masks on
    k = d[names[0]]
    traffic = (2.0 * k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]) * 1024.0
    util = None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k:
        util = k["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / (k["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
    return traffic, util, "profiles/%s: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes" % os.path.basename(PMC_FILE)


def pmc_chamfer(alg_bytes):
    """Counter-measured HBM-side traffic of nn_distance(recon, target) in the loop: chamfer_sym_kernel (column minima resolved inside
    the workgroup, row minima out as per-slice partials) + the part of loss_cgrad_kernel's reads that merges them is NOT split out:
    the scan kernel alone is summed here, and loss_cgrad_kernel's bytes are listed beside it.  (2 * FETCH_SIZE + WRITE_SIZE) KiB per
    launch (gfx950 halves wide reads; separate --pmc passes of tools/attack_breakdown.py 32, i.e. ONE leg: the pruned loop) and the
    profiled kernel times of the same passes.  Dropped when the kernel sources changed."""
    try:
        d = json.load(open(PMC_CHAMFER_FILE))
    except Exception as e:
        return {"traffic": None, "traffic_source": "no PMC profile (%s)" % e}
    want = d.get("_source_sha1", {})
    if not want or source_hashes(sorted(want)) != want:
        return {"traffic": None, "traffic_source": "%s was taken at different kernel sources: dropped" % os.path.basename(PMC_CHAMFER_FILE)}
    tot, us, per = 0.0, 0.0, {}
    for key, counted in (("chamfer_sym_kernel", True), ("loss_cgrad_kernel", False)):
        names = [n for n in d if n.endswith(key)]
        if not names or "FETCH_SIZE" not in d[names[0]] or "WRITE_SIZE" not in d[names[0]]:
            return {"traffic": None, "traffic_source": "%s lacks FETCH_SIZE / WRITE_SIZE for %s" % (os.path.basename(PMC_CHAMFER_FILE), key)}
        k = d[names[0]]
        b = (2.0 * k["FETCH_SIZE"]["mean"] + k["WRITE_SIZE"]["mean"]) * 1024.0
        per[key] = {"bytes_per_launch": b, "avg_us_profiled": k.get("avg_us_profiled")}
        if counted:
            tot += b
            us += k.get("avg_us_profiled") or 0.0
    return {"traffic": tot, "traffic_over_algorithmic": tot / alg_bytes, "achieved_hbm_GBps_counters": (tot / (us * 1e-6) / 1e9) if us else None,
            "per_kernel": per, "traffic_source": "profiles/%s: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch of chamfer_sym_kernel (with the paired "
                                                 "search and the pool Jacobian riding in it), separate --pmc passes of tools/attack_breakdown.py 32 "
                                                 "(the pruned loop only); loss_cgrad_kernel -- which folds the scan's row partials on its way in, "
                                                 "beside its own losses and gradients -- listed, not summed" % os.path.basename(PMC_CHAMFER_FILE)}


def surface_clouds(seed, b, n):
    """Non-uniform stand-ins for ShapeNet surfaces (in_out.py:156-218 loads unit-scaled CAD surfaces; the dataset is absent here):
    60 % of a cloud on a sphere shell of radius 0.4, 35 % on a tilted plane patch through it, 5 % outliers in the unit cube."""
    import numpy as np
    rng = np.random.default_rng(seed)
    k1, k2 = int(0.6 * n), int(0.35 * n)
    v = rng.standard_normal((b, k1, 3))
    shell = 0.4 * v / np.linalg.norm(v, axis=2, keepdims=True)
    uv = rng.random((b, k2, 2)) - 0.5
    plane = np.stack([0.8 * uv[..., 0], 0.8 * uv[..., 1], 0.3 * uv[..., 0] - 0.2 * uv[..., 1] + 0.05], axis=2)
    out = rng.random((b, n - k1 - k2, 3)) - 0.5
    pts = np.concatenate([shell, plane, out], axis=1)
    perm = rng.permuted(np.tile(np.arange(n), (b, 1)), axis=1)                      # no ordering by part
    return np.take_along_axis(pts, perm[:, :, None].repeat(3, 2), axis=1).astype(np.float32)


def surfaces_leg(dev, weights, ae, steps, warmup):
    """secondary.surface_clouds: the headline loop on surface-like clouds instead of uniform cubes, with the paired grid search
    on and off, and how many of the 32 clouds the search hands back to the all-pairs kernel (sampled every 25 iterations)."""
    import torch
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    x, gt = surface_clouds(4001, B, N), surface_clouds(4002, B, N)
    out = {"clouds": "60 % sphere shell r = 0.4, 35 % tilted plane patch, 5 % uniform outliers; seeds 4001 / 4002"}
    for label, prune in (("grid_search", True), ("all_pairs", False)):
        at = AdvAE("adversary", Configuration(batch_size=B, n_points=N, weights=weights, num_iterations=warmup + steps,
                                              num_iterations_thresh=10 ** 6, learning_rate=0.01, chamfer_prune=prune), device=dev, ae=ae)
        at.set_inputs(x, gt, ae.transform(gt), 1.0)
        at.init_pert(None, reset_optimizer=True)
        at.run(0, warmup, 10 ** 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        at.run(warmup, steps, 10 ** 6)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[label] = {"attack_iterations_per_sec": steps / dt, "ms_per_step": dt / steps * 1e3}
        if prune:                       # untimed: the hand-back verdicts along a further stretch of the attack
            hb = []
            for k in range(8):
                at.run(warmup + steps + 25 * k, 25, 10 ** 6)
                hb.append(at.search_state()[1])
            out[label]["clouds_handed_back_of_%d_every_25_iterations" % B] = hb
            out[label]["mean_fraction_handed_back"] = sum(hb) / (len(hb) * float(B))
    return out


class TimedOracle:
    """Oracle proxy that accumulates the seconds spent in the C Chamfer restatement (cpu_baseline's per-part breakdown)."""

    def __init__(self, inner):
        self.inner, self.seconds = inner, 0.0

    def nn_distance(self, a, b):
        t = time.perf_counter()
        out = self.inner.nn_distance(a, b)
        self.seconds += time.perf_counter() - t
        return out


def cpu_baseline(weights, x, gt, iters, gpu_clouds=None):
    """The attack iteration on the host cores, REFERENCE schedule (step = forward + backward + Adam, then a second metrics
    forward: adv_ae.py:217-221), BLAS-backed as SURVEY 8d prescribes: the network in torch-CPU fp32 (oracle/torch_model.py,
    impl "mm": addmm / batch_norm / linear on the MKL GEMMs) + the single-threaded C Chamfer restatement (the
    reference op is single-threaded: NnDistanceOp::Compute, tf_nndistance.cpp:79-80).  Bounded sample: `iters` iterations
    after one warm-up.  A second leg runs the Chamfer restatement under OpenMP too (not what the reference does)."""
    import numpy as np
    import torch
    from geometric_adv_amd.adversary import init_pert_value
    from oracle.cpu_oracle import Oracle
    from oracle.torch_model import TorchAE, TorchAttack
    ae = TorchAE(weights, N, torch.float32, impl="mm")
    # thread count: a [65536, <=256] x [<=256, <=256] GEMM does not scale to every core of a 2-socket host (128 threads
    # ran 8x SLOWER than 8 on the round-2 box); take the best of a few counts on one encoder forward + backward
    xt = torch.as_tensor(x)
    best = (None, 1e30)
    for th in sorted({t for t in (8, 16, 32, 64, torch.get_num_threads()) if t <= (os.cpu_count() or 8)}):
        torch.set_num_threads(th)
        for rep in range(2):
            xr = xt.clone().requires_grad_(True)
            t0 = time.perf_counter()
            ae.encode(xr).sum().backward()
            el = time.perf_counter() - t0
        if el < best[1]:
            best = (th, el)
    threads = best[0]
    torch.set_num_threads(threads)

    def leg(oracle):
        to = TimedOracle(oracle)
        am = TorchAttack(ae, x, gt, None, np.ones(B, np.float32), oracle=to)
        am.init_pert(init_pert_value(B, N))
        am.step(); am.forward()
        to.seconds = 0.0
        t0 = time.perf_counter()
        for _ in range(iters):
            am.step()
            am.forward()
        dt = time.perf_counter() - t0
        return dt, to.seconds

    from oracle.cpu_oracle import Reference
    ref_chamfer = Reference.available()          # oracle/_ref: the reference's OWN nnsearch (tf_nndistance.cpp:21-43) compiled by
    dt, ch = leg(Reference() if ref_chamfer else Oracle())   # oracle/build_ref.sh; it travels to the GPU box as a built .so
    all_cores = None
    try:
        dt2, ch2 = leg(Oracle(omp=True))
        all_cores = {"value": iters / dt2, "sec_per_iteration": dt2 / iters, "chamfer_sec_per_iteration": ch2 / iters,
                     "note": "Chamfer restatement with OpenMP over the clouds of the batch (not what the reference does)"}
    except OSError:                 # OpenMP build of the oracle missing
        pass
    parity = None
    if gpu_clouds is not None:      # the oracle as the checker of the metric's second half: Chamfer rel-err and exact indices
        from geometric_adv_amd import ops
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
    gemm_s = (dt - ch) / iters
    flop = 3 * ENC_FLOP_PER_POINT * B * N + 3 * 2.0 * B * (98304 + 768 * N)      # 2 forwards + 1 backward-to-input
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"value": iters / dt, "unit": "attack-iterations/sec", "cores": threads, "host_cores": os.cpu_count(), "usable_cores": usable,
            "cores_note": "threads of the network's GEMMs = the FASTEST of {8, 16, 32, 64, torch default} <= host cores on one encoder "
                          "forward + backward, measured in this run (not an affinity or cgroup limit: usable_cores says what the process "
                          "may use; [65536 x <=256] x [<=256 x <=256] GEMMs stop scaling long before 256 cores -- 128 threads ran 8 x "
                          "slower than 8 on the round-2 box); the Chamfer part is single-threaded like the reference op",
            "kind": "port",
            "chamfer_kind": "reference (oracle/_ref: the reference's nnsearch compiled from its own source)" if ref_chamfer else "port",
            "parity": parity, "cpu_model": cpu_model,
            "sample": "%d iterations of config 2 (B=32, N=2048) after 1 warm-up, reference schedule (2 forwards + 1 backward per "
                      "iteration); network = torch-CPU %s fp32 (addmm / batch_norm / linear + autograd, best of 8..all threads = %d), Chamfer = "
                      "single-threaded C restatement (gcc -O2 -ffp-contract=off), 4 nn_distance calls per iteration"
                      % (iters, torch.__version__, threads),
            "sec_per_iteration": dt / iters,
            "breakdown_sec_per_iteration": {"network_gemm": gemm_s, "chamfer": ch / iters,
                                            "network_gflops": flop / gemm_s / 1e9},
            "all_cores": all_cores}


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


def emd_leg(dev):
    """Row a15 (config 3's loss): approx_match and the loop's fused levels + cost + gradient at B = 32 and at config 3's per-GPU
    batch B = 128 (N = 2048) against the kernels' bound, which is VALU issue.  Work per pair of points: 21.4 sweep
    pair-weights (10 levels x (B + C + A) sweeps of exp(level * d2) * factor, the first C and the last A missing) at
    3 v_sub + 2 v_mul (2.65 cycles per wave instruction each) + 2 v_fmac_f32 (4.4) + v_exp_f32 (8.3) + v_cvt_f64_f32 +
    v_fma_f64 (4.7 each) = 39.75 cycles per 64 lanes, plus the plan: one distance (19.4 cycles) and 10 x (v_mul, v_exp, v_mul,
    v_fmac) = 10 x 18.0 -- per-opcode issue costs measured on the box (profiles/r01_probe_valu_v2.json), 1024 SIMDs, 2.4 GHz."""
    import torch
    from geometric_adv_amd import ops
    out = {}
    cyc_pw = 3 * 2.65 + 2 * 2.65 + 2 * 4.4 + 8.3 + 4.7 + 4.7
    cyc_pair = 21.4 * cyc_pw + (3 * 2.65 + 2.65 + 2 * 4.4) + 10 * (2.65 + 8.3 + 2.65 + 4.4)
    for b in (32, 128):
        xs, ys = torch.as_tensor(clouds(31, b, N)).to(dev), torch.as_tensor(clouds(32, b, N)).to(dev)

        def timed(f, reps):
            r = f(); r = None; torch.cuda.synchronize()      # (released before the next call: the 2 GB plan of B = 128 is then
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # re-used from the allocator's cache, not malloc'ed in the timed region)
            e0.record()
            for _ in range(reps):
                r = None
                r = f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps, r
        reps = 6 if b == 32 else 3
        t_match, m = timed(lambda: ops.approx_match(xs, ys), reps)
        del m
        t_fused, _ = timed(lambda: ops.emd_cost_grad1(xs, ys), reps)
        bound_ms = b * N * N / 64.0 * cyc_pair / (1024 * 2.4e9) * 1e3
        # round 4: the first three levels' sweeps (8 of the 21.4 pair-weights per pair: A8, B8, C8 + A7, B7, C7 + A6, B6) only meet the
        # pairs inside 3 x 3 x 3 cells of a grid with cells >= the level's reach -- on unit-cube clouds 27 / 16^3, 27 / 12^3, 27 / 6^3
        # of all pairs -- so the work the op actually does is smaller; both fractions are reported
        sparse_pw = 2 * 27 / 16.0 ** 3 + 3 * 27 / 12.0 ** 3 + 3 * 27 / 6.0 ** 3
        bound_reduced_ms = b * N * N / 64.0 * ((21.4 - 8 + sparse_pw) * cyc_pw + cyc_pair - 21.4 * cyc_pw) / (1024 * 2.4e9) * 1e3
        out["B%d" % b] = {"approx_match_ms": t_match, "levels_cost_grad1_fused_ms": t_fused, "issue_bound_ms": bound_ms,
                          "frac": bound_reduced_ms / t_match, "issue_bound_reduced_work_ms": bound_reduced_ms, "frac_reduced_work": bound_reduced_ms / t_match,
                          "equivalent_dense_frac": bound_ms / t_match,
                          "achieved_Tpair_weights_per_s_sweeps_only": 21.4 * b * N * N / (t_match * 1e-3) / 1e12}
    out.update({"bound": "valu issue", "frac_note": "`frac` (= `frac_reduced_work`): issue time of the pair-weights the op actually evaluates -- the sparse "
                "first levels (csrc/emd.hip) skip exact zeros: 8 dense pair-weights per pair become 0.44 at 4-10 x the cost each -- over the "
                "measured time: the roofline fraction.  `equivalent_dense_frac`: the DENSE algorithm's pair-weights over the same time, an "
                "equivalent rate (it counts skipped work), not a roofline fraction", "cycles_per_64_pairs": cyc_pair, "cycles_per_64_sweep_pair_weights": cyc_pw,
                "hbm_note": "approx_match writes the 4 B N M plan once (537 MB at B = 32: ~0.1 ms at 5 TB/s); the fused form writes nothing",
                "round1_approx_match_ms_B32": 4.61, "round3_approx_match_ms_B32": 1.22})
    return out


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


VALU_CYC = 2.65                            # issue cycles of an fp32 add / sub / mul wave instruction (profiles/r01_probe_valu_v2.json)
SIMDS, CLOCK_HZ = 1024, 2.4e9


def _timed(f, reps):
    """ms per call: the median of three windows of `reps` calls after one untimed call (sub-millisecond ops on a box that has just
    run other legs scatter by +- 10 % from window to window)."""
    import torch
    f(); torch.cuda.synchronize()
    w = []
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        w.append((time.perf_counter() - t) / reps * 1e3)
    return sorted(w)[1]


def _timed_attack(at, warm, iters):
    """ms per iteration of an attack handle: `warm` untimed iterations, then `iters` between synchronisations."""
    import torch
    at.run(0, warm, 10 ** 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    at.run(warm, iters, 10 ** 6)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def _timed_attack_windows(at, warm, iters, windows=5, prime_ms=60.0):
    """ms per iteration as the headline measures it: `warm` untimed iterations, ~prime_ms of untimed windows (clock ramp, launch
    queue), then the MEDIAN of `windows` windows of exactly `iters` iterations, each between synchronisations."""
    import torch
    at.run(0, warm, 10 ** 6)
    torch.cuda.synchronize()

    def one(first):
        t0 = time.perf_counter()
        at.run(first, iters, 10 ** 6)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    first = one(warm)
    done = warm + iters
    for _ in range(min(40, int(prime_ms * 1e-3 / max(first, 1e-4)))):
        one(done)
        done += iters
    w = []
    for _ in range(windows):
        w.append(one(done))
        done += iters
    return median(w) / iters * 1e3


def _encoder_frac(at, b, n, first, iters):
    """(avg launch ms, algorithmic fp32 FLOP/s as a fraction of the fp32 MFMA peak -- above 1 is possible: the default arithmetic
    runs on the bf16 pipe, PEAK_ENCODER_X3_TFLOPS = 2.65 x that peak) of the encoder forward from its own begin / end stamps."""
    import torch
    at.profile(["encoder_fwd"], stride=max(1, iters // 32))
    at.run(first, iters, 10 ** 6)
    torch.cuda.synchronize()
    cnt, ms = at.profile_read()["encoder_fwd"]
    at.profile(False)
    avg = ms / max(cnt, 1)
    return avg, ENC_FLOP_PER_POINT * b * n / (avg * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS


def pmc_file_traffic(path, kernels, hashed):
    """(2 * FETCH_SIZE + WRITE_SIZE) KiB per launch summed over `kernels` from a committed tools/pmc_summary.py file, or the
    reason it is not reported (missing, or taken at other kernel sources)."""
    try:
        d = json.load(open(path))
    except Exception as e:
        return None, "no PMC profile (%s)" % e
    want = d.get("_source_sha1", {})
    if not want or source_hashes(sorted(want)) != want or not set(hashed) <= set(want):
        return None, "%s was taken at different kernel sources: dropped" % os.path.basename(path)
    tot = 0.0
    for key in kernels:
        names = [k for k in d if k.endswith(key)]
        if not names or "FETCH_SIZE" not in d[names[0]] or "WRITE_SIZE" not in d[names[0]]:
            return None, "%s lacks FETCH_SIZE / WRITE_SIZE for %s" % (os.path.basename(path), key)
        tot += (2.0 * d[names[0]]["FETCH_SIZE"]["mean"] + d[names[0]]["WRITE_SIZE"]["mean"]) * 1024.0
    return tot, "profiles/%s: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes" % os.path.basename(path)


def config2_leg(dev):
    """BASELINE configs[2]: B = 256, N = 2048, latent-space attack (loss_adv 'latent', distance weight 150: adv_ae.py:107-116) followed
    by the k-NN off-surface defense (defender/get_knn_dists_per_point.py:74-83 + run_defense_surface.py:187-207)."""
    import torch
    from geometric_adv_amd import defense, ops, weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    b = 256
    w = W.synthetic_weights(N, seed=7)
    ae = PointNetAE(w, N, device=dev)
    x, gt = clouds(1003, b, N), clouds(2003, b, N)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=N, weights=w, loss_adv_type="latent", loss_dist_type="chamfer",
                                          dist_weight_list=[150.0], num_iterations=60, num_iterations_thresh=10 ** 6), device=dev, ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 150.0)
    at.init_pert(None, reset_optimizer=True)
    ms_it = _timed_attack(at, 5, 20)
    enc_ms, enc_frac = _encoder_frac(at, b, N, 25, 16)
    adv = at.peek()["adv"]
    xs = torch.as_tensor(x).to(dev)
    t_knn = _timed(lambda: ops.knn_dists(adv, 8), 5)
    t_knn_point = _timed(lambda: ops.knn_point(9, adv, adv), 3)
    t_def = _timed(lambda: defense.defend_surface_device(ae, adv, xs), 5)
    t_def_np = _timed(lambda: defense.defend_surface(ae, adv, xs), 2)
    pairs = float(b) * N * N
    bound_ms = pairs / 64.0 * 8 * VALU_CYC / (SIMDS * CLOCK_HZ) * 1e3
    out = {"workload": "BASELINE configs[2]: B=256 x N=2048, latent attack (dist weight 150) + kNN defense (k=8, 2-NN mean > 0.04)",
           "attack_ms_per_iteration": ms_it, "attack_iterations_per_sec": 1e3 / ms_it,
           "encoder_fwd": {"avg_launch_ms": enc_ms, "frac_of_fp32_mfma_peak": enc_frac},
           "knn_dists_k8_ms": t_knn, "knn_point_k9_ms": t_knn_point,
           "defend_surface_ms": t_def, "defend_surface_numpy_in_out_ms": t_def_np,
           "roofline_knn": {"bound": "valu issue", "kernel": "knn_grid_build_kernel + knn_grid_kernel<1, 9, 512> (exact grid search: lane-private 27-cell walk, leftovers by a cooperative all-points scan) + knn_redo_kernel for tied / non-finite queries",
                            "pairs_per_launch": pairs, "achieved_Tpair_per_s": pairs / (t_knn * 1e-3) / 1e12,
                            "issue_bound_ms": bound_ms, "equivalent_all_pairs_frac": bound_ms / t_knn,
                            "pairs_evaluated_per_query": KNN_CANDIDATES_PER_QUERY,
                            "frac": bound_ms * (KNN_CANDIDATES_PER_QUERY / N) / t_knn,
                            "bound_note": "distance only: 8 fp32 VALU instructions per pair (3 sub, 3 mul, 2 add -- unfused, the reference's "
                                          "rounding) at %.2f issue cycles per wave instruction, %d SIMDs, %.1f GHz.  `frac`: the distance "
                                          "evaluations the grid search PERFORMS (~%d candidates per query on uniform clouds: its own 27 cells, "
                                          "tools/debug/knn_diag.py) against that bound -- the search is bound by its walk and list bookkeeping, "
                                          "not by distance arithmetic; `equivalent_all_pairs_frac`: all b*n*n pairs over the same time, an "
                                          "equivalent rate, not a roofline fraction" % (VALU_CYC, SIMDS, CLOCK_HZ / 1e9, KNN_CANDIDATES_PER_QUERY),
                            "algorithmic_bytes_per_launch": 12.0 * b * N + 4.0 * b * N * 8}}
    del at
    return out


def config3_leg(dev, emd):
    """BASELINE configs[3] per GPU: B = 128 (1024 over 8 GPUs), N = 2048, loss_adv = Chamfer + approx-EMD / N (SURVEY a15;
    tf_approxmatch.cpp:23-84)."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    b = 128
    w = W.synthetic_weights(N, seed=7)
    ae = PointNetAE(w, N, device=dev)
    x, gt = clouds(1004, b, N), clouds(2004, b, N)
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=N, weights=w, num_iterations=20, num_iterations_thresh=10 ** 6,
                                          emd_weight=1.0), device=dev, ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0)
    at.init_pert(None, reset_optimizer=True)
    ms_it = _timed_attack(at, 2, 6)
    e = (emd or {}).get("B128", {})
    out = {"workload": "BASELINE configs[3] per GPU: B=128 x N=2048, loss_adv = Chamfer + approx-EMD/N (lambda 1), dist chamfer w=1",
           "attack_ms_per_iteration": ms_it, "attack_iterations_per_sec": 1e3 / ms_it,
           "emd_levels_cost_grad1_fused_ms": e.get("levels_cost_grad1_fused_ms"),
           "emd_share_of_iteration": (e.get("levels_cost_grad1_fused_ms") / ms_it) if e.get("levels_cost_grad1_fused_ms") else None,
           "roofline_emd": {"bound": "valu issue", "approx_match_ms": e.get("approx_match_ms"), "issue_bound_ms": e.get("issue_bound_ms"),
                            "frac": e.get("frac"), "equivalent_dense_frac": e.get("equivalent_dense_frac"), "see": "secondary.roofline_emd"}}
    del at
    out["full_size_ms_per_iteration"] = _full_size_ms(dev, 1024, N, 1.0)
    out["full_size_note"] = "configs[3] WHOLE on this GPU: B=1024 x N=2048, Chamfer + approx-EMD/N, 3 timed iterations after 2"
    return out


def _full_size_ms(dev, b, n, emd_weight, warm=2, iters=3):
    """ms per attack iteration of a whole multi-GPU config of BASELINE.json on THIS GPU (the N = 1 point a measured 8-GPU figure
    would be divided by; parity of the same run: tests/test_gpu_configs.py::test_config*_full_size_*)."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    w = W.synthetic_weights(n, seed=7)
    ae = PointNetAE(w, n, device=dev)
    x, gt = clouds(1006, b, n), clouds(2006, b, n)
    kw = {"emd_weight": emd_weight} if emd_weight else {}
    at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=warm + iters + 1,
                                          num_iterations_thresh=10 ** 6, **kw), device=dev, ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0)
    at.init_pert(None, reset_optimizer=True)
    ms = _timed_attack(at, warm, iters)
    at.status()
    del at, ae
    return ms


KNN_CANDIDATES_PER_QUERY = 110      # measured by tools/debug/knn_diag.py on uniform clouds of 2048 points (knn row)
PMC_CHAMFER_8192 = os.path.join(ROOT, "profiles", "r06_pmc_chamfer_n8192.json")
MX_MFMA_CYC, MX_MIN_CYC, MX_MIN_PER_PAIR = 32.0, 4.75, 1.19     # screened scan: one 32x32x16 fp16 MFMA (8 passes) per 1024 pairs; min-class VALU
                                                                  # instructions (4.75 issue cycles, profiles/r01_probe_valu_v2.json) per pair: 0.5 row + 0.69 column


def config4_leg(dev):
    """BASELINE configs[4] per GPU: B = 32 (256 over 8 GPUs), N = 8192 dense clouds, output-space attack (chamfer / chamfer)."""
    from geometric_adv_amd import weights as W
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    b, n = 32, 8192
    w = W.synthetic_weights(n, seed=7)
    ae = PointNetAE(w, n, device=dev)
    x, gt = clouds(1005, b, n), clouds(2005, b, n)
    out = {"workload": "BASELINE configs[4] per GPU: B=32 x N=8192, output-space attack (chamfer/chamfer, w=1)"}
    for label, prune in (("grid_search", True), ("all_pairs", False)):
        at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=100, num_iterations_thresh=10 ** 6,
                                              chamfer_prune=prune), device=dev, ae=ae)
        at.set_inputs(x, gt, ae.transform(gt), 1.0)
        at.init_pert(None, reset_optimizer=True)
        ms_it = _timed_attack(at, 4, 16)
        out[label] = {"attack_ms_per_iteration": ms_it, "attack_iterations_per_sec": 1e3 / ms_it}
        if prune:
            enc_ms, enc_frac = _encoder_frac(at, b, n, 20, 16)
            at.profile(True)
            at.run(36, 10, 10 ** 6)
            br = {k: (ms / max(c_, 1)) for k, (c_, ms) in at.profile_read().items()}
            at.profile(False)
            ch_ms = br["chamfer_fwd"]
            alg = 20.0 * b * (n + n)
            from geometric_adv_amd import _lib
            screened = _lib.lib().geoadv_nn_distance_sym_is_screened(b // 4, n, n) == 1     # (the loop's rule, four tiles per CU = the operator's rule at a quarter of the batch)
            traffic, note = pmc_file_traffic(PMC_CHAMFER_8192, ("chamfer_mx_kernel<8>" if screened else "chamfer_sym_kernel", "chamfer_sym_merge_kernel"),
                                             ("chamfer_sym.hip", "chamfer_mx.h"))
            pairs = 2.0 * b * n * n
            bound_ms = (pairs / 2) / 64.0 * 8 * VALU_CYC / (SIMDS * CLOCK_HZ) * 1e3     # one distance evaluation serves both directions
            # the screened kernel's own issue bound: per 64 pairs 1/16 of an MFMA and 1.19 min-class VALU instructions
            bound_mx_ms = (pairs / 2) / 64.0 * (MX_MFMA_CYC / 16.0 + MX_MIN_PER_PAIR * MX_MIN_CYC) / (SIMDS * CLOCK_HZ) * 1e3
            out["encoder_fwd"] = {"avg_launch_ms": enc_ms, "frac_of_fp32_mfma_peak": enc_frac}
            out["kernel_ms_per_iteration"] = br
            out["roofline_chamfer"] = {"bound": "valu issue", "kernel": ("chamfer_mx_kernel<8> (matrix-pipe-screened scan, csrc/chamfer_mx.h)" if screened else "chamfer_sym_kernel") +
                                       " + chamfer_sym_merge_kernel: nn_distance(recon, target) at N = 8192 (the paired search for nn_distance(adv, x) in a "
                                       "launch of its own at this size)", "screened": screened,
                                       "scan_plus_finish_class_ms": ch_ms, "achieved_Tpair_per_s": pairs / (ch_ms * 1e-3) / 1e12,
                                       "issue_bound_ms": bound_mx_ms if screened else bound_ms, "frac": (bound_mx_ms if screened else bound_ms) / ch_ms,
                                       "unscreened_issue_bound_ms": bound_ms, "equivalent_unscreened_frac": bound_ms / ch_ms,
                                       "bound_note": "screened: approximate distances from one fp16 MFMA per 32 x 32 pairs (32 pipe cycles) and 1.19 min-class "
                                                     "VALU instructions per pair (4.75 issue cycles each) -- the work this kernel does on EVERY pair; the exact "
                                                     "evaluations of the selected ~3 % are overhead against it.  `equivalent_unscreened_frac`: the unscreened "
                                                     "scan's bound (8 fp32 VALU instructions per distance, once for both directions) over the same time: an "
                                                     "equivalent rate, not a roofline fraction" if screened else
                                                     "8 fp32 VALU instructions per distance, every distance evaluated once for both directions",
                                       "algorithmic_bytes_per_launch": alg, "traffic": traffic, "traffic_source": note,
                                       "traffic_over_algorithmic": (traffic / alg) if traffic else None}
        del at
    out["full_size_ms_per_iteration"] = _full_size_ms(dev, 256, n, 0.0)
    out["full_size_note"] = "configs[4] WHOLE on this GPU: B=256 x N=8192, output-space attack, 3 timed iterations after 2"
    return out


def victim_shapes(rng, count, n):
    """Ellipsoid and box surfaces of random aspect (tools/trained_victim_attack.py): what the stand-in victim is trained on."""
    import numpy as np
    u = rng.standard_normal((count, n, 3)).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    scale = rng.uniform(0.15, 0.45, size=(count, 1, 3)).astype(np.float32)
    box = rng.random((count, 1, 1)) < 0.5
    return (np.where(box, np.clip(u * 3.0, -1.0, 1.0), u) * scale).astype(np.float32)


def trained_victim_leg(dev, steps, warmup, train_steps=320):
    """The headline loop against a TRAINED victim (the reference attacks a trained AE: attacker/run_attack.py:80-137 with the
    model of autoencoder/train_ae.py:43-79; the headline's victim has random-init weights).  The AE is trained here, untimed and
    seeded, with the repo's own trainer on synthetic surfaces; then the standard window runs with the paired grid search on, off,
    and in the adaptive default (off once most of the batch is handed back), with the hand-back rate along the attack and an
    index-parity check of nn_distance(adv, x) against the pinned oracle on 4 clouds."""
    import numpy as np
    import torch
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    from geometric_adv_amd.trainer import PointNetAETrainer, initial_weights
    rng = np.random.default_rng(0)
    tr = PointNetAETrainer(initial_weights(N, seed=2), N, batch_size=50, learning_rate=0.001, device=dev)
    data = victim_shapes(rng, 400, N)
    loss = None
    for ep in range(train_steps // 8):
        loss, _ = tr._single_epoch_train(data)
    w = tr.export_weights()
    del tr
    ae = PointNetAE(w, N, device=dev)
    src, tgt = victim_shapes(rng, B, N), victim_shapes(rng, B, N)
    out = {"victim": "this package's trainer, %d steps of batch 50 on 400 synthetic ellipsoid / box surfaces (seed 0), lr 0.001; "
                     "final epoch loss %.5f" % (train_steps // 8 * 8, float(loss)),
           "attack": "B=32 x N=2048, chamfer/chamfer, w=1, lr 0.01, %d warm-up iterations, then the headline's protocol: ~60 ms of "
                     "untimed windows, median of 5 windows of %d iterations" % (warmup, steps)}
    for label, prune in (("grid_search", True), ("all_pairs", False)):
        at = AdvAE("adversary", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=warmup + steps,
                                              num_iterations_thresh=10 ** 6, learning_rate=0.01, chamfer_prune=prune), device=dev, ae=ae)
        at.set_inputs(src, tgt, ae.transform(tgt), 1.0)
        at.init_pert(None, reset_optimizer=True)
        hb = []
        if prune:                                    # untimed first pass: the verdicts along the attack
            for k in range(8):
                at.run(k * (warmup + steps) // 8, (warmup + steps) // 8, 10 ** 6)
                hb.append(at.search_state()[1])
            at.init_pert(None, reset_optimizer=True)
        ms_it = _timed_attack_windows(at, warmup, steps)
        out[label] = {"attack_iterations_per_sec": 1e3 / ms_it, "ms_per_step": ms_it}
        if prune:
            out[label]["clouds_handed_back_of_%d_along_the_attack" % B] = hb
            out["mean_fraction_handed_back"] = sum(hb) / (len(hb) * float(B))
            handed = at.adapt_source_search()        # the default policy's decision at the end of a run (AdvAE._attack_one_batch)
            out["adaptive_default"] = {"search_switched_off": not getattr(at, "_search_on", True), "clouds_handed_back_at_decision": handed}
            if not getattr(at, "_search_on", True):
                at.init_pert(None, reset_optimizer=True)
                ms_ad = _timed_attack_windows(at, warmup, steps)
                out["adaptive_default"].update({"attack_iterations_per_sec": 1e3 / ms_ad, "ms_per_step": ms_ad})
            # parity of what the search answered (or handed to the all-pairs kernel) on the moved clouds, against the oracle
            from oracle.cpu_oracle import Oracle
            p = at.peek()
            adv4 = p["adv"][:4].cpu().numpy()
            _, oi1, _, oi2 = Oracle().nn_distance(adv4, src[:4])
            out["index_parity_4_clouds"] = bool(np.array_equal(p["idx_a1"][:4].cpu().numpy(), oi1) and
                                                np.array_equal(p["idx_a2"][:4].cpu().numpy(), oi2))
            pn = p["pert"].norm(dim=2).flatten()
            out["pert_norm_median"] = float(pn.median().item())
            out["pert_norm_p99"] = float(torch.quantile(pn[:1000000], 0.99).item())
        del at
    return out


class Leg:
    """One attack handle on this rank + the timed-window protocol."""

    def __init__(self, dev, weights, ae, x, gt, warmup, steps, prune=True, dog=None):
        import torch
        self.dog = dog
        from geometric_adv_amd.adv_ae import AdvAE, Configuration
        self.K, self.W = steps, warmup
        self.thresh = warmup + int(0.8 * steps) + 1
        b = x.shape[0]
        conf = Configuration(batch_size=b, n_points=N, weights=weights, loss_adv_type="chamfer", loss_dist_type="chamfer",
                             dist_weight_list=[1.0], num_iterations=warmup + steps, num_iterations_thresh=self.thresh,
                             learning_rate=0.01, chamfer_prune=prune)
        self.ref = torch.as_tensor(ae.get_loss_per_pc(gt)).to(dev)      # target_ae_loss_ref
        self.at = AdvAE("adversary", conf, device=dev, ae=ae)
        self.at.set_inputs(x, gt, ae.transform(gt), 1.0)
        self.at.init_pert(None, reset_optimizer=True)
        self.at.run(0, warmup, self.thresh)                             # W untimed warm-up steps
        self.gathered = None

    def window(self, gdist, backend, dev):
        """Exactly K timed steps bracketed by barrier + synchronize on both sides; returns the max over ranks (seconds)."""
        import torch
        if self.dog:
            self.dog.arm("a timed window of %d steps (barrier, loop, gather of the final scalars, barrier)" % self.K)
        torch.cuda.synchronize()
        gdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.at.run(self.W, self.K, self.thresh)                        # no host sync inside
        metrics, _, _ = self.at.get_best(self.ref, clouds=False)      # the final loss scalars (the clouds stay where they are)
        self.gathered = gdist.all_gather_examples(metrics[None] if backend == "nccl" else metrics[None].cpu(), axis=1)   # final loss scalars only
        torch.cuda.synchronize()
        gdist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dt = gdist.max_over_ranks(dt, device=dev if backend == "nccl" else "cpu")
        if self.dog:
            self.dog.disarm()
        return dt

    def windows(self, count, gdist, backend, dev, prime_ms=100.0):
        """`count` timed windows after untimed priming windows worth ~prime_ms of GPU work (an idle chip ramps its clock over
        the first tens of ms: a 20-step window is 4 ms)."""
        first = self.window(gdist, backend, dev)
        for _ in range(min(40, int(prime_ms * 1e-3 / max(first, 1e-4)))):
            self.window(gdist, backend, dev)
        return [self.window(gdist, backend, dev) for _ in range(count)]


def rccl_selftest_child():
    """One-rank 'nccl' (= RCCL) group on cuda:0: barrier, the metrics all-gather, a sum all-reduce and the max-over-ranks of the
    timing protocol on device tensors.  RCCL refuses two ranks on one device; one rank it accepts, so a 1-GPU box can show that
    librccl loads, the communicator comes up and the collectives of the N > 1 path complete."""
    import torch
    from geometric_adv_amd import dist as gdist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    t0 = time.perf_counter()
    gdist.init("nccl", single_rank_group=True, timeout_s=60)
    dev = torch.device("cuda", 0)
    gdist.barrier()
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    m = torch.arange(32 * 5, dtype=torch.float32, device=dev).reshape(1, 32, 5)
    ok = bool(torch.equal(gdist.all_gather_examples(m), m))
    t = torch.ones(1 << 20, device=dev)
    gdist.all_reduce_sum_(t)
    ok = ok and bool((t == 1).all().item()) and gdist.max_over_ranks(1.5, device=dev) == 1.5
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        gdist.all_gather_examples(m)
    torch.cuda.synchronize()
    info = gdist.backend_info()
    info.update({"ok": ok, "communicator_up_s": round(t_up, 3), "metrics_gather_us": round((time.perf_counter() - t1) / 20 * 1e6, 1)})
    import torch.distributed as tdist
    tdist.destroy_process_group()
    print("RCCL_SELFTEST " + json.dumps(info))


def rccl_selftest():
    """Runs rccl_selftest_child in a CHILD process (bounded by a timeout; this process keeps the GPU state of the finished
    measurement) and returns its report, or the reason it failed."""
    try:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-selftest"], capture_output=True, text=True, timeout=180,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": "timed out after 180 s"}
    for ln in o.stdout.splitlines():
        if ln.startswith("RCCL_SELFTEST "):
            return json.loads(ln[len("RCCL_SELFTEST "):])
    return {"ok": False, "error": (o.stderr or o.stdout)[-400:]}


def small_batch_sweep(dev, weights, ae, warmup, steps, gdist, backend, dog):
    """ms per iteration of ONE GPU at B = 16 / 8 / 4 (rank 0's global batch, first B clouds), measured in this run with the
    headline's own window protocol: what each rank of ONE batch of 32 split over 2 / 4 / 8 GPUs executes (the
    reference's default batch is 10, attacker/run_attack.py:36).  Windows of >= 50 steps (>= 4 ms), median of 5."""
    out = {}
    k = max(steps, 50)
    for bs in SWEEP_BATCHES:
        leg = Leg(dev, weights, ae, clouds(1002, B, N)[:bs], clouds(2002, B, N)[:bs], warmup, k, dog=dog)
        dts = leg.windows(5, gdist, backend, dev, prime_ms=20.0)
        out[bs] = {"ms_per_step": median(dts) / k * 1e3, "windows_ms": [round(t * 1e3, 3) for t in dts], "steps_per_window": k}
        del leg
    return out


def calibration(dev):
    """What THIS box delivers to three bare probes, outside every timed window (<= 50 ms of GPU time): the fp32 MFMA issue rate
    (tools/probe: 32x32x2 f32, four independent accumulators, no memory traffic), the issue cost of an fp32 VALU multiply
    (SIMD cycles per wave instruction at the nominal 2.4 GHz) and a 256 MB device-to-device copy.  A headline that moves with
    these moved with the box, not with the kernels."""
    import torch
    out = {}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools", "probe"))
        import probe
        iters = 200
        ms = min(probe.microbench(7, iters) for _ in range(3))
        out["mfma_f32_probe_tflops"] = 4096 * 4 * 16 * iters * (2 * 32 * 32 * 2) / ms / 1e9
        iters = 1000
        ms = min(probe.microbench(19, iters) for _ in range(3))
        out["valu_probe_cycles_per_instr"] = ms * 1e-3 * CLOCK_HZ * SIMDS / (2048 * 4 * 16 * iters)
    except Exception as e:                  # measurement tooling only: its absence must not fail the bench
        out["probe_error"] = str(e)[-200:]
    n = 64 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b_ = torch.empty_like(a)
    b_.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b_.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    out["hbm_copy_GBps"] = 2 * 4.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out["note"] = ("tools/probe kernels 7 (v_mfma_f32_32x32x2_f32, 4 accumulators; peak %.1f) and 19 (v_mul_f32; cycles per wave "
                   "instruction at %.1f GHz, %d SIMDs), torch copy_ of 256 MB (read + write bytes); outside the timed windows"
                   % (PEAK_MFMA_F32_TFLOPS, CLOCK_HZ / 1e9, SIMDS))
    return out


def sustained_bf16_tflops():
    """TFLOP/s a bare stream of the encoder's MFMAs (the f16x2 loop shape) sustained in the recorded probe (tools/bf16x3_probe.py
    f16x2, weights from LDS, 3 VALU per MFMA), or None when the record is missing: a number of ANOTHER box and clock state,
    reported as such."""
    try:
        rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r06_f16x2_probe.jsonl"))]
        v = [r["f16_tflops"] for r in rows if r.get("probe") == "throughput_f16x2" and r.get("valu_per_mfma") == 3 and not r.get("gap_kernel_ms")]
        return sum(v) / len(v) if v else None
    except Exception:
        return None


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def main():
    args = parse_args()
    if args.rccl_selftest:
        return rccl_selftest_child()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)                                    # never returns
    if "WORLD_SIZE" not in os.environ:
        ensure_built()

    import numpy as np
    import torch
    from geometric_adv_amd import _lib, dist as gdist
    rank, world, local = gdist.env_rank()
    if not os.path.exists(_lib.LIB_PATH):
        raise SystemExit("bench.py: %s is missing; under an external torchrun build first "
                         "(python -c 'import __graft_entry__ as g; g.build()')" % _lib.LIB_PATH)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # GEOADV_BENCH_SHARE_GPU=1: every rank on cuda:0 with a gloo group -- the 2-ranks-on-one-GPU simulation of the N > 1
    # path (tests/test_gpu_configs.py); RCCL refuses two ranks on one device
    share = os.environ.get("GEOADV_BENCH_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count()
    local = 0 if share else local
    if local >= ndev:
        raise SystemExit("bench.py: rank %d wants cuda:%d but the node has %d GPUs" % (rank, local, ndev))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ONE backend for every rank, fixed before the first collective: RCCL (one rank per GPU), or gloo for the shared-GPU
    # simulation.  There is no per-rank fallback -- ranks on different backends would wait for each other for ever; a group
    # that cannot come up ends the run with a message (watchdog), non-zero.
    backend = "gloo" if share else "nccl"
    dog = Watchdog(rank, args.window_timeout)
    if world > 1:
        dog.arm("process-group start-up (%s) + first barrier" % ("RCCL" if backend == "nccl" else backend))
        gdist.init(backend, timeout_s=args.window_timeout)
        gdist.barrier()
        if backend == "nccl":
            torch.cuda.synchronize()
        dog.disarm()
    group_info = gdist.backend_info()

    from geometric_adv_amd import weights as W
    from geometric_adv_amd.autoencoder import PointNetAE

    K, Wm, R = args.steps, args.warmup, max(1, args.windows)
    weights = W.synthetic_weights(N, seed=7)
    ae = PointNetAE(weights, N, device=dev)
    if args.only_leg:
        leg_out = {"config2": lambda: config2_leg(dev), "config3": lambda: config3_leg(dev, emd_leg(dev)), "config4": lambda: config4_leg(dev),
                   "trained_victim": lambda: trained_victim_leg(dev, min(K, 200), Wm), "training": lambda: training_leg(dev),
                   "emd": lambda: emd_leg(dev), "surfaces": lambda: surfaces_leg(dev, weights, ae, min(K, 200), Wm)}[args.only_leg]()
        print(json.dumps({"leg": args.only_leg, **leg_out}))
        return
    x = clouds(1000 + 2 + 17 * rank, B, N)            # source batch of this rank
    gt = clouds(2000 + 2 + 17 * rank, B, N)           # target batch of this rank

    # ---- headline leg: one batch of 32 per rank (weak scaling), paired grid search on ----
    leg = Leg(dev, weights, ae, x, gt, Wm, K, dog=dog)
    # kernel-timed encoder launches: ~64 of them over the timed windows (>= 50 even at the driver's --steps 20).  A stamped
    # launch costs the loop ~5 us (measured: --steps 20 with every launch stamped 5100 it/s, every 4th 5207), so they are
    # kept sparse; the event pool holds 2048 pairs
    stride = max(1, (R * K + 1499) // 1500, (R * K) // 64)
    leg.windows(0, gdist, backend, dev)                        # priming only
    leg.at.profile(["encoder_fwd"], stride=stride)             # the kernel's own begin / end stamps (hipExtLaunchKernel): no extra
    dts = leg.windows(R, gdist, backend, dev, prime_ms=0.0)    # packets between dependent kernels, agrees with rocprofv3
    prof = leg.at.profile_read()
    leg.at.profile(False)
    dt = median(dts)

    # ---- the same with nn_distance(adv, x) by the all-pairs kernel (no data-dependent shortcut) ----
    leg_ap = Leg(dev, weights, ae, x, gt, Wm, K, prune=False, dog=dog)
    dts_ap = leg_ap.windows(min(R, 3), gdist, backend, dev)
    dt_ap = median(dts_ap)
    del leg_ap

    # ---- the same loop with the encoder's products as fp32 MFMAs (GEOADV_ENC_ARITH_F32) / as six bf16 piece products
    # (GEOADV_ENC_ARITH_BF16X3, round 5's default): what the default arithmetic buys ----
    dt_f32 = dt_f32ap = dt_x3 = None
    if not args.no_secondary:                   # (--no-secondary: the counter passes profile the headline launches only)
        ae_x3 = PointNetAE(weights, N, device=dev, encoder_arith="bf16x3")
        leg_x3 = Leg(dev, weights, ae_x3, x, gt, Wm, K, dog=dog)
        dt_x3 = median(leg_x3.windows(min(R, 3), gdist, backend, dev))
        del leg_x3, ae_x3
        ae_f32 = PointNetAE(weights, N, device=dev, encoder_arith="f32")
        leg_f32 = Leg(dev, weights, ae_f32, x, gt, Wm, K, dog=dog)
        dts_f32 = leg_f32.windows(min(R, 3), gdist, backend, dev)
        dt_f32 = median(dts_f32)
        del leg_f32
        # ---- neither shortcut: fp32 MFMA encoder AND all-pairs nn_distance(adv, x) ----
        leg_f32ap = Leg(dev, weights, ae_f32, x, gt, Wm, K, prune=False, dog=dog)
        dt_f32ap = median(leg_f32ap.windows(min(R, 3), gdist, backend, dev))
        del leg_f32ap, ae_f32

    # ---- strong scaling: ONE global batch of 32 (rank 0's seeds), 32 / world clouds per rank ----
    strong = None
    if B % world == 0:
        bs = B // world
        if world == 1:
            strong = {"value": K / dt, "ms_per_step": dt / K * 1e3, "note": "one GPU: identical to `value`"}
        else:
            xs, gs = clouds(1002, B, N)[rank * bs:(rank + 1) * bs], clouds(2002, B, N)[rank * bs:(rank + 1) * bs]
            leg_s = Leg(dev, weights, ae, xs, gs, Wm, K, dog=dog)
            dts_s = leg_s.windows(R, gdist, backend, dev)
            dt_s = median(dts_s)
            strong = {"value": K / dt_s, "ms_per_step": dt_s / K * 1e3, "windows_ms": [round(t * 1e3, 3) for t in dts_s]}
            del leg_s
        strong.update({"definition": "attack iterations/s on ONE global batch of 32 clouds split contiguously over the ranks "
                                     "(SURVEY 8e); every rank runs the whole loop on its 32/N clouds, final scalars all-gathered",
                       "global_batch": B, "batch_per_gpu": bs})
        if world == 1 and not args.no_secondary:       # (--no-secondary: the counter passes profile the headline launches only)
            # what a rank of the 2 / 4 / 8-GPU strong-scaled run executes, TIMED HERE (not a table): B = 16 / 8 / 4 on this GPU
            sweep = small_batch_sweep(dev, weights, ae, Wm, K, gdist, backend, dog)
            ms32 = dt / K * 1e3
            strong["measured_ms"] = {"32": ms32, **{str(k): v["ms_per_step"] for k, v in sweep.items()}}
            strong["measured_windows"] = {str(k): v for k, v in sweep.items()}
            strong["projected_speedup_at_gpus"] = {str(B // k): ms32 / v["ms_per_step"] for k, v in sweep.items()}
            strong["projection_note"] = ("ms per iteration of THIS GPU at B = 32 / 16 / 8 / 4, timed in this run with the headline's window "
                                         "protocol; projected speed-up of one batch of 32 over G GPUs = ms(32) / ms(32 / G) (no data-path "
                                         "collective; the final gather of scalars is outside the loop)")

    if world > 1:                                # every collective of the run is behind us: leave the group cleanly on all ranks
        import torch.distributed as tdist
        gdist.barrier()
        if tdist.is_initialized():
            tdist.destroy_process_group()
    if rank != 0:
        return
    at = leg.at
    total = Wm + K
    # per-kernel-class breakdown in a separate, untimed pass (bracketing events: each class interval includes its dispatch gaps)
    at.profile(True)
    at.run(total, 50, total + 1000)
    torch.cuda.synchronize()
    breakdown = {k: (ms / max(n_, 1)) for k, (n_, ms) in at.profile_read().items()}
    at.profile(False)

    enc_n, enc_ms = prof["encoder_fwd"]
    enc_avg_ms = enc_ms / max(enc_n, 1)
    enc_flop = ENC_FLOP_PER_POINT * B * N                       # algorithmic flop per launch
    enc_tflops = enc_flop / (enc_avg_ms * 1e-3) / 1e12
    traffic, mfma_util, traffic_note = pmc_encoder()
    ch_avg_ms = breakdown["chamfer_fwd"]
    ch_pairs = 2.0 * B * N * N                                  # nn_distance(recon, target), 2 directions per step (adv/x: grid search)
    ch_bytes = 20.0 * B * (N + N)                               # 20*B*(N+M) per nn_distance call
    out = {
        "metric": "attack-iterations/sec (B=32, N=2048) at 1/2/4/8 GPUs; Chamfer rel-err vs ref",
        "value": world * K / dt, "unit": "attack-iterations/sec", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (encoder products: f16x2)", "data": "synthetic",
        "encoder_arith": "f16x2 (fp32 operands, scaled by powers of two, as two fp16 pieces; three piece products per multiply, fp32 "
                         "accumulate; range-guarded; include/geoadv.h GEOADV_ENC_ARITH_F16X2) -- everything else plain fp32",
        "timing": "median of %d windows of exactly %d steps, each bracketed by barrier + synchronize, max over ranks; ~0.1 s of "
                  "untimed priming windows first (clock ramp)" % (R, K),
        "windows_ms": [round(t * 1e3, 3) for t in dts],
        "value_all_pairs": world * K / dt_ap,
        "value_all_pairs_note": "nn_distance(adv, x) by the all-pairs kernel for every cloud (Configuration.chamfer_prune=False): what a "
                                "victim whose adversarial points leave their grid cells gets; same results bit for bit",
        "value_encoder_f32": (world * K / dt_f32) if dt_f32 else None,
        "value_encoder_f32_note": "the same loop with the encoder's products as fp32 MFMAs (encoder_arith='f32', v_mfma_f32_32x32x2_f32: "
                                  "the round-4 kernel).  `value` runs the default: every fp32 operand as two fp16 pieces of its power-of-two-"
                                  "scaled value, a product as its three piece products of weight >= 2^-11 on v_mfma_f32_32x32x16_f16, fp32 "
                                  "accumulate -- the error of an fp32 accumulation in another order (profiles/r06_f16x2_probe.jsonl: rms "
                                  "0.44-0.52 against 0.43-0.46 units of 2^-24 |a|.|w|), same parity tolerances (tests/test_gpu_encoder_x3.py)",
        "value_encoder_bf16x3": (world * K / dt_x3) if dt_x3 else None,
        "value_encoder_bf16x3_note": "the same loop with round 5's default arithmetic (three bf16 pieces, six piece products): the fallback of a "
                                     "model whose activations leave f16x2's range (>= 1023.5; guarded, never silent)",
        "config": {"workload": "BASELINE configs[1]: B=32 random clouds x N=2048, output-space attack (chamfer/chamfer, "
                               "dist_weight 1.0, lr 0.01), one batch per GPU", "batch_per_gpu": B, "n_points": N,
                   "global_batch": B * world, "parallelism": "batches sharded, dp%d, no data-path collective" % world,
                   "ranks": world, "collective_backend": group_info["backend"], "collective_world": group_info["world"],
                   "rccl_version": group_info["rccl_version"],
                   "thresh_fraction": 0.8},
        "strong_scaling": strong,
        "roofline": {"bound": "mfma", "kernel": "encoder_fwd3_kernel<2, true>", "achieved": enc_tflops, "peak": PEAK_ENCODER_X3_TFLOPS,
                     "unit": "TFLOP/s", "frac": enc_tflops / PEAK_ENCODER_X3_TFLOPS, "traffic": traffic, "traffic_source": traffic_note,
                     "peak_note": "algorithmic fp32 FLOP (SURVEY 8d: 2 x 90304 per point) against the dense fp16 MFMA peak (%.0f TFLOP/s) / %d: "
                                  "the kernel issues three fp16 piece products per fp32 product.  Against the fp32 MFMA peak the round-4 kernel "
                                  "was priced on (%.1f) the same launch is frac_of_fp32_mfma_peak" % (PEAK_MFMA_BF16_TFLOPS, X3_PRODUCTS, PEAK_MFMA_F32_TFLOPS),
                     "issued_f16_tflops": X3_PRODUCTS * enc_tflops, "frac_of_fp32_mfma_peak": enc_tflops / PEAK_MFMA_F32_TFLOPS,
                     "sustained_note": "a bare stream of the same MFMAs in the kernel's loop shape (operands in registers, weight pieces re-read "
                                       "from LDS, 3 VALU per MFMA, all 256 CUs, random operands) is what frac_of_sustained_f16_stream compares "
                                       "with (profiles/r06_f16x2_probe.jsonl, another box): the chip lowers its clock under matrix load "
                                       "(MI355X_MICROARCH.md, DVFS give-back; the bf16 stream of round 5 sustained 1.37 PFLOP/s = 55 % of the spec peak)",
                     "frac_of_sustained_f16_stream": (X3_PRODUCTS * enc_tflops / sustained_bf16_tflops()) if sustained_bf16_tflops() else None,
                     "mfma_pipe_utilisation_pmc": mfma_util,
                     "avg_launch_ms": enc_avg_ms, "launches_timed": enc_n, "algorithmic_flop_per_launch": enc_flop,
                     "timing": "kernel begin/end stamps (hipExtLaunchKernel start/stop events) of every %s launch inside the "
                               "timed windows" % ("" if stride == 1 else {2: "2nd", 3: "3rd"}.get(stride, "%d-th" % stride))},
        "roofline_chamfer": {"bound": "valu", "kernel": "chamfer_sym_kernel: nn_distance(recon, target), both directions from one distance "
                                                         "evaluation per pair, column minima and indices resolved inside the workgroup, row "
                                                         "minima merged by the loss launch (no second Chamfer launch); nn_distance(adv, x) is "
                                                         "answered exactly by the paired grid search, whose workgroups ride in the scan's launch "
                                                         "beside the encoder's pool Jacobian",
                             "avg_class_ms": ch_avg_ms, "launches_timed": 50,
                             "issue_bound_ms": (ch_pairs / 2) / 64.0 * 8 * VALU_CYC / (SIMDS * CLOCK_HZ) * 1e3,
                             "frac": (ch_pairs / 2) / 64.0 * 8 * VALU_CYC / (SIMDS * CLOCK_HZ) * 1e3 / ch_avg_ms,
                             "bound_note": "headline shape (B = 32, N = 2048): 8 unfused fp32 VALU instructions per distance (the reference's "
                                           "rounding), every distance evaluated once for both directions, %.2f issue cycles per wave instruction, "
                                           "%d SIMDs, %.1f GHz; the class time carries the two riders too" % (VALU_CYC, SIMDS, CLOCK_HZ / 1e9),
                             "achieved_Tpair_per_s": ch_pairs / (ch_avg_ms * 1e-3) / 1e12,
                             "algorithmic_bytes_per_launch": ch_bytes,
                             "achieved_hbm_GBps": ch_bytes / (ch_avg_ms * 1e-3) / 1e9, "hbm_peak_GBps": PEAK_HBM_GBS,
                             **pmc_chamfer(ch_bytes)},
        "kernel_ms_per_iteration": breakdown,
        "kernel_ms_note": "bracketing events per class (dispatch gaps included), untimed 50-iteration pass; encoder_fwd here is kernel-timed",
        "final_mean_target_recon_error": float(leg.gathered[0, :, 4].mean().item()),
    }
    out["calibration"] = calibration(dev)
    hb_headline = leg.at.search_state()
    out["paired_search"] = {"in_use": hb_headline[0], "clouds_handed_back_of_%d_at_the_end" % B: hb_headline[1],
                            "note": "`value` has nn_distance(adv, x) answered by the exact paired grid search; on this victim (random-init "
                                    "weights, as the metric's config prescribes) no cloud is handed back.  A trained victim moves points "
                                    "out of the grid cells: secondary.trained_victim reports what it gets (= value_all_pairs' path; the "
                                    "default policy switches the search off by itself)"}
    if world == 1 and not args.no_rccl_selftest:   # no collective ran at N = 1: show in a child process that RCCL comes up on this box
        out["config"]["rccl_selftest"] = rccl_selftest()
    if world == 1 and not args.no_secondary:       # the other BASELINE configs and the widened rows, beside the headline (not part of `value`)
        emd = emd_leg(dev)
        out["secondary"] = {"configs": {"config2_latent_knn_b256": config2_leg(dev), "config3_chamfer_emd_b128": config3_leg(dev, emd),
                                        "config4_n8192_b32": config4_leg(dev)},
                            "trained_victim": trained_victim_leg(dev, min(K, 200), Wm),
                            "ae_training_step": training_leg(dev), "roofline_emd": emd,
                            "surface_clouds": surfaces_leg(dev, weights, ae, min(K, 200), Wm)}
        profiled = any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ)
        slot_counts = [args.slots] if args.slots > 1 else ([2, 3] if args.slots < 0 and not profiled else [])
        if slot_counts:
            out["secondary"]["batch_slots"] = {str(c): slots_leg(dev, weights, ae, c) for c in slot_counts}
            out["secondary"]["batch_slots"]["note"] = ("Configuration.batch_slots: that many INDEPENDENT B = 32 batches of an attack set in flight on this GPU "
                                                       "(own handle, stream and host thread each): one batch's dependent launches and their boundaries run under "
                                                       "another's kernels.  Aggregate iterations/s over the slots; `value` stays one batch at a time")
    if world == 1 and not args.no_cpu_baseline:
        _, adv_best, recon_best = at.get_best(leg.ref)
        out["cpu_baseline"] = cpu_baseline(weights, x, gt, args.cpu_iters,
                                           gpu_clouds=(recon_best[:4].contiguous(), torch.as_tensor(gt[:4]).to(dev)))
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    # The qualifiers of the headline, repeated (short) under `config` and `roofline`: the driver's record keeps only
    # metric ... config, roofline and cpu_baseline of this line (VERDICT r05 task 2).
    rc = out["roofline_chamfer"]
    sec = out.get("secondary") or {}
    tv = sec.get("trained_victim") or {}
    r1 = lambda v: None if v is None else round(float(v), 1)
    out["config"]["qualifiers"] = {
        "value_all_pairs": r1(out["value_all_pairs"]), "value_encoder_f32": r1(out["value_encoder_f32"]),
        "value_encoder_bf16x3": r1(out["value_encoder_bf16x3"]),
        "value_f32_all_pairs": r1((world * K / dt_f32ap) if dt_f32ap else None),
        "trained_victim_its": {k: r1(tv[k]["attack_iterations_per_sec"]) for k in ("grid_search", "all_pairs", "adaptive_default")
                               if isinstance(tv.get(k), dict) and "attack_iterations_per_sec" in tv[k]} or None,
        "strong_measured_ms": ({k: round(v, 4) for k, v in strong["measured_ms"].items()} if strong and "measured_ms" in strong else None),
        "batch_slots_its": {k: r1(v["attack_iterations_per_sec_all_slots"]) for k, v in (sec.get("batch_slots") or {}).items() if isinstance(v, dict)} or None,
        "full_size_ms": {k: (sec.get("configs", {}).get(k) or {}).get("full_size_ms_per_iteration")
                         for k in ("config3_chamfer_emd_b128", "config4_n8192_b32")} if sec else None,
        "note": "all_pairs: nn_distance(adv, x) without the data-dependent grid search; encoder_f32 / encoder_bf16x3: fp32 MFMAs / six bf16 "
                "piece products instead of f16x2's three fp16 ones; "
                "f32_all_pairs: neither; strong_measured_ms: ms per iteration of THIS GPU at B = 32/G; batch_slots_its: aggregate it/s with "
                "2 / 3 independent B = 32 batches in flight (not the headline's one batch at a time); full_size_ms: configs[3] "
                "(B=1024) and configs[4] (B=256 x 8192) whole on one GPU"}
    out["roofline"]["chamfer"] = {"kernel": "chamfer scan of nn_distance(recon, target)", "frac": round(rc["frac"], 4),
                                  "avg_class_ms": round(rc["avg_class_ms"], 5), "issue_bound_ms": round(rc["issue_bound_ms"], 5),
                                  "traffic_over_algorithmic": rc.get("traffic_over_algorithmic")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
