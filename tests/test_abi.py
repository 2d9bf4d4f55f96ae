"""CPU: the C-ABI library loads and exports every symbol include/geoadv.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "geoadv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(geoadv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ["geoadv_nn_distance", "geoadv_nn_distance_grad", "geoadv_approx_match", "geoadv_match_cost",
                 "geoadv_match_cost_grad", "geoadv_selection_sort", "geoadv_group_point", "geoadv_group_point_grad",
                 "geoadv_query_ball_point", "geoadv_knn_point", "geoadv_attack_run"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  -- before the library, so that both bind to one HIP runtime if GPU tests share the process
    from geometric_adv_amd import _lib
    import __graft_entry__
    if not os.path.exists(_lib.LIB_PATH):
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, "libgeoadv.so lacks: %s" % missing
    assert lib.geoadv_version() >= 1


def _struct_fields(name):
    """(C type, field) pairs of `typedef struct <name> { ... } <name>;` in include/geoadv.h, comments stripped."""
    text = open(os.path.join(ROOT, "include", "geoadv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), text, flags=re.S).group(1)
    return [(t, f) for t, f in re.findall(r"\b(int|float)\s+([a-z0-9_]+)\s*;", body)]


def test_python_mirror_of_the_attack_config_matches_the_header():
    """The ctypes structure adv_ae.py passes to geoadv_attack_create has the header's fields, in order, with the header's types:
    a field added on one side only would shift every later one silently."""
    from geometric_adv_amd.adv_ae import _AttackConfig
    want = _struct_fields("geoadv_attack_config")
    got = [({ctypes.c_int: "int", ctypes.c_float: "float"}[t], f) for f, t in _AttackConfig._fields_]
    assert got == want
    assert want[-1] == ("int", "loss_in_scan")


def test_python_mirror_of_the_train_config_matches_the_header():
    from geometric_adv_amd.trainer import _TrainConfig
    want = _struct_fields("geoadv_train_config")
    got = [({ctypes.c_int: "int", ctypes.c_float: "float"}[t], f) for f, t in _TrainConfig._fields_]
    assert got == want
