"""nn_distance through the symmetric scan ALONE (no riders), B = 32 x 2048 x 2048 and 32 x 8192 x 8192: the subject of the counter
passes that price the scan's own HBM-side traffic (tools/collect_pmc.sh).   python tools/debug/sym_only.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geometric_adv_amd import ops
for b, n in ((32, 2048),):
    x = torch.rand((b, n, 3), device="cuda") - 0.5
    y = torch.rand((b, n, 3), device="cuda") - 0.5
    for _ in range(40):
        ops.nn_distance_sym(x, y)
    torch.cuda.synchronize()
