"""ms per iteration with the symmetric scan forced on / off around the small-batch threshold (attack.hip: chamfer_sym).
    python tools/debug/sym_threshold.py [B ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = "import sys; sys.path.insert(0, %r + '/tools'); import attack_sweep as s, json; print(json.dumps(s.run(int(sys.argv[1]), 2048, 300, **json.loads(sys.argv[2]))))" % ROOT
for B in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 24]:
    row = {"batch": B}
    for i, val in enumerate(['two_scan', 'symmetric']):
        out = subprocess.run([sys.executable, "-c", code, str(B), json.dumps({'chamfer_kernel': val})], capture_output=True, text=True).stdout
        row["sym" + str(i)] = round(json.loads(out.strip().splitlines()[-1])["ms_per_iteration"], 4)
    print(json.dumps(row), flush=True)
