"""Per-launch durations of ONE fused approx-EMD call (levels + cost + gradient) from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/debug/emd_trace.py run ; python3 tools/debug/emd_trace.py show OUT"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "run":
    import numpy as np, torch
    from geometric_adv_amd import ops
    B, N = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 2048
    rng = np.random.default_rng(B)
    blob = len(sys.argv) > 3 and sys.argv[3] == "blob"          # cloud 1 = a Gaussian blob (std 0.022) inside the unit cloud 2: the attack's random-init reconstruction
    x = torch.as_tensor((rng.standard_normal((B, N, 3)) * 0.022).astype(np.float32) if blob else rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    y = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    for _ in range(3):
        ops.emd_cost_grad1(x, y)
    torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "emd" in r["Kernel_Name"]]
    per = len(rows) // 3
    last = rows[-per:]
    tot = 0.0
    for r in last:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        print("%-60s %8.1f us" % (r["Kernel_Name"].split("(")[0].replace("void geoadv::", "")[:60], d))
    print("sum of kernels %.1f us; span %.1f us" % (tot, (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
