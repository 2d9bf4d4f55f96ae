"""nn_distance(P, Q) at the loop's shapes: public op (two scans), symmetric scan + finish (chamfer_sym.hip, through the bulk scorer's
entry point is not comparable -- so timed via the attack loop's breakdown instead) and the packed symmetric kernel.
    python tools/chamfer_pk_time.py [B ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_adv_amd import ops
N = int(os.environ.get("GEOADV_TOOL_N", "2048"))


def timed(f, reps=50):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32, 64]:
    rng = np.random.default_rng(B)
    x = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    y = torch.as_tensor(rng.random((B, N, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    print(json.dumps({"batch": B, "n": N, "two_scan_op_us": round(timed(lambda: ops.nn_distance(x, y)), 2),
                      "packed_symmetric_us_incl_fill_unpack": round(timed(lambda: ops.nn_distance_symmetric(x, y)), 2)}), flush=True)
