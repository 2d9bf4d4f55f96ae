"""nn_distance through the symmetric scan ALONE (operator form, no riders): the subject of the counter passes that price the scan's
own instruction count and HBM-side traffic (tools/collect_pmc.sh).
    python tools/debug/sym_only.py [unscreened|screened] [N]        default: unscreened 2048 (the B = 32 loop's kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geometric_adv_amd import ops
screened = len(sys.argv) > 1 and sys.argv[1] == "screened"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ops.chamfer_screen(screened)
b = 32
x = torch.rand((b, n, 3), device="cuda") - 0.5
y = torch.rand((b, n, 3), device="cuda") - 0.5
for _ in range(40 if n <= 2048 else 8):
    ops.nn_distance_sym(x, y)
torch.cuda.synchronize()
