import sys, os, json
sys.path.insert(0, "/root/repo/tools")
import attack_sweep as s
for B, N, it in [(32, 8192, 40), (256, 2048, 60), (32, 2048, 300), (32, 4096, 100)]:
    print(json.dumps(s.run(B, N, it)))
