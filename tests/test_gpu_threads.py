"""GPU: the kernel-selection switches are per call / per handle (SURVEY 8b threading: a re-entrant library with no globals on the
launch path) -- two host threads run DIFFERENT selections concurrently and each gets the bits of its own serial run."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_knn_and_emd_selections_per_call_from_two_host_threads():
    import torch
    from geometric_adv_amd import ops
    from geometric_adv_amd.adv_ae import AdvAE, Configuration
    from geometric_adv_amd import weights as W
    from conftest import cloud
    dev = torch.device("cuda:0")
    pc = torch.as_tensor(cloud(71, 48, 2048)).to(dev)
    q = torch.as_tensor(cloud(72, 48, 1500)).to(dev)
    x, y = torch.as_tensor(cloud(73, 6, 1024)).to(dev), torch.as_tensor(cloud(74, 6, 1024)).to(dev)
    # serial references, one per selection
    want = {}
    for kern in ("all_points", "grid", "grid_shells"):
        want[kern] = (ops.knn_dists(pc, 8, kernel=kern), ops.knn_point(9, pc, q, kernel=kern))
    for kern in ("grid", "grid_shells"):                      # the kernels agree bit for bit (same results, include/geoadv.h)
        assert torch.equal(want[kern][0], want["all_points"][0])
        assert torch.equal(want[kern][1][0], want["all_points"][1][0]) and torch.equal(want[kern][1][1], want["all_points"][1][1])
    emd = {dense: (ops.approx_match(x, y, dense_levels=dense), ops.emd_cost_grad1(x, y, dense_levels=dense)) for dense in (False, True)}
    torch.cuda.synchronize()
    errors = []

    def work(kern, dense, reps):
        try:
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                for _ in range(reps):
                    d = ops.knn_dists(pc, 8, kernel=kern)
                    v, i = ops.knn_point(9, pc, q, kernel=kern)
                    m = ops.approx_match(x, y, dense_levels=dense)
                    c, g = ops.emd_cost_grad1(x, y, dense_levels=dense)
                    torch.cuda.current_stream().synchronize()
                    ok = (torch.equal(d, want[kern][0]) and torch.equal(v, want[kern][1][0]) and torch.equal(i, want[kern][1][1])
                          and torch.equal(m, emd[dense][0]) and torch.equal(c, emd[dense][1][0]) and torch.equal(g, emd[dense][1][1]))
                    if not ok:
                        errors.append((kern, dense))
                        return
        except BaseException as e:          # surfaced in the main thread
            errors.append(e)

    ts = [threading.Thread(target=work, args=("all_points", True, 12)), threading.Thread(target=work, args=("grid", False, 12))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    # per handle: two attacks with Chamfer + EMD loss, one with dense EMD sweeps, side by side == each alone
    n, b = 1024, 4
    w = W.synthetic_weights(n, seed=3)
    xs, gs = cloud(75, b, n), cloud(76, b, n)

    def attack(dense, out, stream=None):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            at = AdvAE("adversary", Configuration(batch_size=b, n_points=n, weights=w, num_iterations=6, num_iterations_thresh=1,
                                                  emd_weight=1.0, emd_dense_levels=dense))
            at.set_inputs(xs, gs, None, 1.0)
            at.init_pert(None, reset_optimizer=True)
            hist = torch.empty((6, 6, b), device=dev)
            at.run(0, 6, 1, hist)
            torch.cuda.current_stream().synchronize()
            out[dense] = hist.cpu().numpy()

    alone, together = {}, {}
    attack(False, alone)
    attack(True, alone)
    ts = [threading.Thread(target=attack, args=(d, together, torch.cuda.Stream(dev))) for d in (False, True)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for d in (False, True):
        assert np.array_equal(alone[d], together[d]), d
    np.testing.assert_allclose(alone[False], alone[True], rtol=1e-5)


def test_workspace_forms_refuse_small_workspaces():
    import ctypes as C
    import torch
    from geometric_adv_amd import _lib
    from conftest import cloud
    dev = torch.device("cuda:0")
    pc = torch.as_tensor(cloud(5, 2, 600)).to(dev)
    out = torch.empty((2, 600, 4), device=dev)
    need = int(_lib.lib().geoadv_knn_workspace_bytes(2, 600, 600, 5))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    st = _lib.lib().geoadv_knn_dists_ws(0, 2, 600, 4, _lib.ptr(pc), _lib.ptr(out), _lib.ptr(ws), C.c_size_t(need - 1), _lib.stream_handle())
    assert st == 1 and b"workspace too small" in _lib.lib().geoadv_last_error()
    st = _lib.lib().geoadv_knn_dists_ws(7, 2, 600, 4, _lib.ptr(pc), _lib.ptr(out), _lib.ptr(ws), C.c_size_t(need), _lib.stream_handle())
    assert st == 1
    st = _lib.lib().geoadv_knn_dists_ws(2, 2, 600, 4, _lib.ptr(pc), _lib.ptr(out), _lib.ptr(ws), C.c_size_t(need), _lib.stream_handle())
    assert st == 0
