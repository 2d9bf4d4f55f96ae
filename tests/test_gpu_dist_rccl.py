"""GPU: RCCL really comes up on the box.  tests/test_dist_gloo.py covers the N > 1 logic on CPU; here the same helpers
(barrier, all_gather_examples, all_reduce_sum_, max_over_ranks, attack_sharded) run on DEVICE tensors through a one-rank
'nccl' group -- the first RCCL initialisation in this repository happens in a test, not on the driver's 8-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_group_collectives():
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    o = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in o.stdout.splitlines() if ln.startswith("RCCL_CHILD ")]
    assert o.returncode == 0 and lines, o.stderr[-1500:]
    r = json.loads(lines[-1][len("RCCL_CHILD "):])
    assert r["gather_equal"] and r["sum_ok"] and r["max"] == 7.25 and r["attack_equal"] and r["slice"] == [0, 4]
    assert r["info"]["backend"] == "rccl" and r["info"]["world"] == 1 and r["info"]["rccl_version"]
