"""ms per iteration at the batch sizes the latency work is judged on: python tools/debug/sweep_small.py [B ...]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import attack_sweep as s
row = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32]:
    best = min(s.run(B, 2048, 300)["ms_per_iteration"] for _ in range(2))
    row[str(B)] = round(best, 4)
print(json.dumps(row))
