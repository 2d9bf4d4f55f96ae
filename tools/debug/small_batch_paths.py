"""ms per iteration of the small batches on the path choices attack.hip picks between (two scans + masked backward /
symmetric scan + riders + pool Jacobian; paired grid search or all pairs for nn_distance(adv, x)).
    python tools/debug/small_batch_paths.py [B ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = "import sys; sys.path.insert(0, %r + '/tools'); import attack_sweep as s, json; print(json.dumps(s.run(int(sys.argv[1]), 2048, 300, **json.loads(sys.argv[2]))))" % ROOT
CFGS = {"auto": {}, "two_scan": {"chamfer_kernel": "two_scan"}, "two_scan+grid": {"chamfer_kernel": "two_scan", "chamfer_prune": "always"},
        "sym": {"chamfer_kernel": "symmetric"}, "sym+grid": {"chamfer_kernel": "symmetric", "chamfer_prune": "always"},
        "sym+allpairs": {"chamfer_kernel": "symmetric", "chamfer_prune": False}}
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5]:
    row = {"batch": B}
    for name, cfg in CFGS.items():
        out = subprocess.run([sys.executable, "-c", code, str(B), json.dumps(cfg)], capture_output=True, text=True)
        try:
            row[name] = round(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_iteration"], 4)
        except Exception:
            row[name] = out.stderr[-300:]
    print(json.dumps(row), flush=True)
