"""Launch-by-launch durations of the EMD part of ONE Chamfer + EMD attack iteration (last of 6) from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/debug/emd_attack_trace.py run B SPARSE ; ... show OUT"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "run":
    import numpy as np, torch
    from geometric_adv_amd import weights as W, ops
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd.autoencoder import PointNetAE
    N, B = 2048, int(sys.argv[2])
    ops.emd_sparse_levels(sys.argv[3] == "1")
    rng = np.random.default_rng(B)
    x = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5); gt = rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)
    w = W.synthetic_weights(N, seed=7); ae = PointNetAE(w, N)
    at = AdvAE("a", Configuration(batch_size=B, n_points=N, weights=w, num_iterations=400, num_iterations_thresh=10**6, emd_weight=1.0), ae=ae)
    at.set_inputs(x, gt, ae.transform(gt), 1.0); at.init_pert(None, reset_optimizer=True)
    at.run(0, 6, 10**6); torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "emd_init_kernel" in r["Kernel_Name"]]
    last = rows[starts[-1]:]
    end = next(i for i, r in enumerate(last) if "emd_cost_fold" in r["Kernel_Name"])
    t0 = int(last[0]["Start_Timestamp"])
    for r in last[:end + 1]:
        print("%-42s start %8.1f  dur %8.1f us" % (r["Kernel_Name"].split("(")[0].replace("void geoadv::", "").replace("geoadv::", "")[:42],
                                                   (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
