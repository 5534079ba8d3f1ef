"""CPU checks of the BVH oracle beyond the reference's three hand-built trees: the memory-layout invariant
the reference's helper verifies (tests/test_helpers/ptbvhtree_helper.gd:60-94: an inner node's children, and a
leaf's objects, occupy consecutive slots starting at its index) on every tree type the oracle builds, the
binned-SAH rule's own promises, and the refit's update_aabb semantics (pt_bvh_tree.gd:254-302,873-882) on
cases whose outcome can be stated without running the reference."""
import numpy as np
import pytest

import oracle_binding as ob
from test_oracle_golden import _decode

SAH = 5


def _tree(pta, scene, bvh_type, order=8):
    wd = pta.work_dispatcher
    for m in scene.meshes:
        m.bvh_type = bvh_type
    nodes, ids = wd.create_scene_bvh(scene, ob.OracleBvhBuilder(), order, bvh_type)
    return _decode(nodes), np.asarray(ids)


def _check_layout(nodes, ids, order, n_objects):
    """Every node but the root is a child of exactly one node; every id slot belongs to exactly one leaf; sizes
    respect the order; a parent's box holds its children's (sockets excepted: their child lives in mesh space)."""
    parent_of = np.full(len(nodes), -1)
    slot_owner = np.full(len(ids), -1)
    for i, n in enumerate(nodes):
        assert 0 <= n["size"] <= order, f"node {i} holds {n['size']} entries"
        if n["index"] > 0:  # inner node or socket: children at index .. index + size - 1
            for c in range(n["index"], n["index"] + n["size"]):
                assert parent_of[c] == -1, f"node {c} has two parents"
                parent_of[c] = i
                # (a leaf of degenerate objects only keeps the zero box: merge_aabb ignores invalid boxes, pt_utils.gd:34-43)
                if n["xform"] < 0 and any(nodes[c]["min"]) | any(nodes[c]["max"]):
                    assert all(nodes[c]["min"][a] >= n["min"][a] - 1e-4 for a in range(3))
                    assert all(nodes[c]["max"][a] <= n["max"][a] + 1e-4 for a in range(3))
        elif n["index"] < 0:  # leaf: id slots -index-1 .. -index-1 + size - 1
            first = -n["index"] - 1
            for s in range(first, first + n["size"]):
                assert slot_owner[s] == -1, f"id slot {s} belongs to two leaves"
                slot_owner[s] = i
    assert (parent_of[1:] >= 0).all() and parent_of[0] == -1, "a node is unreachable"
    assert (slot_owner >= 0).all(), "an id slot belongs to no leaf"
    assert len(ids) == n_objects and len(set(ids.tolist())) == n_objects, "every object exactly once"


def _count(scene):
    return len(scene.spheres) + len(scene.triangles) + sum(len(m.triangle_set) for m in scene.meshes if m.triangle_set is not None)


@pytest.mark.parametrize("name", ["sphere_scene1", "benchmark_random_scene", "fake_cornell_box", "dragon_test", "test_scene_bvh_b"])
@pytest.mark.parametrize("bvh_type", [1, 2, 3, 4, SAH])
def test_layout_invariant_every_type(pta, golden_dir, name, bvh_type):
    scene = pta.scenes.load_fixture_scene(name, golden_dir)
    nodes, ids = _tree(pta, scene, bvh_type)
    _check_layout(nodes, ids, 8, len(ids))
    assert len(ids) >= len(scene.spheres)


@pytest.mark.parametrize("order", [2, 3, 4, 8])
def test_layout_invariant_orders(pta, golden_dir, order):
    scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
    nodes, ids = _tree(pta, scene, 4, order)
    _check_layout(nodes, ids, order, len(ids))
    if order != 8:  # the SAH rule is defined for 8-wide nodes only
        with pytest.raises(RuntimeError):
            _tree(pta, scene, SAH, order)


def _sah_cost(nodes):
    """Sum over inner nodes of (children's half-areas / own half-area): the quantity the binned rule lowers."""
    def area(n):
        d = [max(n["max"][a] - n["min"][a], 0.0) for a in range(3)]
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
    total = 0.0
    for n in nodes:
        if n["index"] > 0 and n["xform"] < 0 and area(n) > 0:
            total += sum(area(nodes[c]) for c in range(n["index"], n["index"] + n["size"])) / area(n)
    return total


def test_sah_tree_is_cheaper_than_the_sorted_tree(pta):
    """Same objects, same order: the SAH tree's summed relative child area is well below the longest-axis tree's
    on clustered input (the property the builder exists for); both satisfy the layout invariant."""
    scene = pta.scenes.atrium_scene(16384, seed=7)
    n_sorted, ids_a = _tree(pta, scene, 4)
    n_sah, ids_b = _tree(pta, scene, SAH)
    _check_layout(n_sorted, ids_a, 8, len(ids_a))
    _check_layout(n_sah, ids_b, 8, len(ids_b))
    assert sorted(ids_a.tolist()) == sorted(ids_b.tolist())
    assert _sah_cost(n_sah) < 0.8 * _sah_cost(n_sorted), (_sah_cost(n_sah), _sah_cost(n_sorted))


# ---- refit ---------------------------------------------------------------------------------------------
def _sphere_box(center, r):
    c = np.asarray(center, np.float32)
    r = np.float32(r)
    return np.concatenate([c - r, np.array([r + r] * 3, np.float32)]).astype(np.float32)  # pt_sphere.gd:58-65


def _refit_setup(pta, golden_dir):
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
    oracle = ob.OracleBvhBuilder()
    tree, _ = wd.build_scene_tree(scene, oracle, 8)
    nodes, ids = oracle.read(tree)
    sph = np.frombuffer(wd.create_spheres_byte_array(scene), np.float32).reshape(-1, 8)
    return oracle, tree, bytearray(nodes), np.asarray(ids), sph, len(scene.spheres)


def _event(i, center, r):
    v = np.zeros(12, np.float32)
    v[:6] = _sphere_box(center, r)
    return i | (1 << 24), v


def _leaf_of(nodes, ids, object_id):
    slot = int(np.nonzero(ids == object_id)[0][0])
    for i, n in enumerate(nodes):
        if n["index"] < 0 and -n["index"] - 1 <= slot < -n["index"] - 1 + n["size"]:
            return i
    raise AssertionError("no leaf holds the object")


def _parents(nodes):
    parent = {}
    for i, n in enumerate(nodes):
        if n["index"] > 0:
            for c in range(n["index"], n["index"] + n["size"]):
                parent[c] = i
    return parent


def test_refit_with_unchanged_boxes_queues_nothing(pta, golden_dir):
    oracle, tree, nodes, ids, sph, n = _refit_setup(pta, golden_dir)
    before = bytes(nodes)
    evs = [_event(i, sph[i, :3], sph[i, 3]) for i in (0, 3, 7, n - 1)]
    queued = oracle.refit(tree, [e[0] for e in evs], [e[1] for e in evs], nodes)
    assert len(queued) == 0 and bytes(nodes) == before
    oracle.free(tree)


def test_refit_queue_is_a_path_towards_the_root(pta, golden_dir):
    """update_aabb climbs from the object's leaf while the box keeps changing (pt_bvh_tree.gd:873-882): the queue
    starts at the leaf, each entry is the parent of the one before, and the rewritten boxes hold the new sphere."""
    oracle, tree, nodes, ids, sph, n = _refit_setup(pta, golden_dir)
    dec0 = _decode(bytes(nodes))
    parent = _parents(dec0)
    i = 5
    center = sph[i, :3] + np.array([40.0, 0.0, 0.0], np.float32)  # far outside the scene: every ancestor grows
    oid, v = _event(i, center, sph[i, 3])
    queued = oracle.refit(tree, [oid], [v], nodes).tolist()
    leaf = _leaf_of(dec0, ids, oid)
    assert queued[0] == leaf
    for a, b in zip(queued, queued[1:]):
        assert parent[a] == b
    assert queued[-1] == 0, "a move this large reaches the root"
    dec1 = _decode(bytes(nodes))
    for q in queued:
        assert dec1[q]["max"][0] >= center[0] + sph[i, 3] - 1e-3
        assert dec1[q]["node_id"] == dec0[q]["node_id"] and dec1[q]["xform"] == dec0[q]["xform"], "topology fields untouched"
    untouched = [k for k in range(len(dec0)) if k not in queued]
    assert all(dec1[k] == dec0[k] for k in untouched)
    oracle.free(tree)


def test_refit_there_and_back_restores_the_bytes(pta, golden_dir):
    oracle, tree, nodes, ids, sph, n = _refit_setup(pta, golden_dir)
    before = bytes(nodes)
    i = 11
    oid, away = _event(i, sph[i, :3] + np.array([0.0, 25.0, 0.0], np.float32), sph[i, 3])
    _, home = _event(i, sph[i, :3], sph[i, 3])
    q1 = oracle.refit(tree, [oid], [away], nodes)
    assert len(q1) > 0 and bytes(nodes) != before
    q2 = oracle.refit(tree, [oid], [home], nodes)
    assert q2.tolist() == q1.tolist() and bytes(nodes) == before
    oracle.free(tree)


def test_refit_below_tolerance_stops_at_the_leaf(pta, golden_dir):
    """A move far below is_equal_approx's 1e-5 leaves the leaf's box 'equal': nothing is queued, yet the object's
    own box is replaced, so a later real move of a leaf-mate starts from it."""
    oracle, tree, nodes, ids, sph, n = _refit_setup(pta, golden_dir)
    before = bytes(nodes)
    i = 2
    oid, tiny = _event(i, sph[i, :3] + np.array([2e-7, 0.0, 0.0], np.float32), sph[i, 3])
    assert len(oracle.refit(tree, [oid], [tiny], nodes)) == 0 and bytes(nodes) == before
    with pytest.raises(RuntimeError):
        oracle.refit(tree, [(1 << 24) | 0x7FFFFF], [tiny], nodes)  # unknown object
    oracle.free(tree)
