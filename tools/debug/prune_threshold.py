"""ms per iteration with the paired grid search on / off at small batches (attack.hip: chamfer_prune).
    python tools/debug/prune_threshold.py [B ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = "import sys; sys.path.insert(0, %r + '/tools'); import attack_sweep as s, json; print(json.dumps(s.run(int(sys.argv[1]), 2048, 300, **json.loads(sys.argv[2]))))" % ROOT
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]:
    row = {"batch": B}
    for i, val in enumerate([False, True]):
        out = subprocess.run([sys.executable, "-c", code, str(B), json.dumps({'chamfer_prune': val})], capture_output=True, text=True).stdout
        row["prune" + str(i)] = round(json.loads(out.strip().splitlines()[-1])["ms_per_iteration"], 4)
    print(json.dumps(row), flush=True)
