"""Work plan: the reference's own tests (ryhope/src/storage/updatetree.rs:545-632) restated over the
C ABI, plus the ordering invariant they rely on (children before parents). CPU only: host logic."""
import importlib

import pytest

mp2 = importlib.import_module("mapreduce-plonky2_amd")
wp = importlib.import_module("mapreduce-plonky2_amd.workplan")

PATHS = [[1, 3, 57, 9, 0], [1, 3, 89, 20], [1, 3, 57, 9, 10], [1, 3, 57, 43, 1874]]  # updatetree.rs:549-554


def make_tree():
    mt = wp.UpdateTree.from_path(PATHS[0], 3)
    for p in PATHS[1:]:
        mt.extend_with_path(p)
    return mt


def test_tree_shape():
    mt = make_tree()
    assert len(mt) == 10 and mt.epoch == 3 and mt.root() == 1
    assert mt.nodes() == [1, 3, 57, 9, 0, 89, 20, 10, 43, 1874]  # arena = insertion order
    par = mt.parents()
    assert par[1] is None and par[3] == 1 and par[57] == 3 and par[89] == 3 and par[9] == 57 and par[43] == 57
    assert par[0] == 9 and par[10] == 9 and par[20] == 89 and par[1874] == 43
    assert mt.path_ends() == {0, 20, 10, 1874}
    assert mt.subtree_size(57) == 6 and mt.subtree_size(1) == 10 and mt.subtree_size(20) == 1
    assert mt.contains_key(43) and not mt.contains_key(44)
    assert wp.UpdateTree.from_paths(PATHS, 3).nodes() == mt.nodes()
    assert len(wp.UpdateTree.from_paths([], 0)) == 0


def test_mt_creation():
    """updatetree.rs:545-582: batch size 1 = leaf-first traversal in waves"""
    plan = make_tree().into_workplan()
    parents = plan.tree().parents()
    finished, waves = set(), []
    while True:
        wave = wp.drain_wave(plan)
        for it in wave:
            assert it.subtree is None
            kids = [k for k, p in parents.items() if p == it.k]
            assert all(k in finished for k in kids), "a parent was handed out before its children"
        for it in wave:
            plan.done(it.k)
            finished.add(it.k)
        if not wave:
            break
        waves.append([it.k for it in wave])
    assert plan.completed() and finished == set(parents)
    assert waves[0] == [1874, 10, 20, 0]  # anchors are consumed last-in first-out
    assert waves[-1] == [1]
    assert [it for w in waves for it in w if it in (0, 20, 10, 1874)] == [1874, 10, 20, 0]


def test_path_end_flags():
    plan = make_tree().into_workplan()
    ends = {}
    while True:
        wave = wp.drain_wave(plan)
        if not wave:
            break
        for it in wave:
            ends[it.k] = it.as_node()
            plan.done(it.k)
    assert {k for k, e in ends.items() if e} == {0, 20, 10, 1874}


def test_mt_creation_staggered():
    """updatetree.rs:584-632: every batch size from 0 to size+1 terminates and yields every node once"""
    n = len(make_tree())
    for batch_size in range(0, n + 2):
        plan = make_tree().into_batched_workplan(batch_size)
        count_done, seen = 0, []
        while True:
            nxt = plan.next()
            if nxt is None or not nxt.ready:
                break
            item = nxt.item
            if item.subtree is not None:
                count_done += len(item.subtree)
                seen += item.subtree.nodes()
                assert item.subtree.root() == item.k
                if batch_size >= 1:
                    assert len(item.subtree) <= max(batch_size, 1)
                order = item.subtree.bottom_up()
                par = item.subtree.parents()
                pos = {k: i for i, k in enumerate(order)}
                assert all(p is None or pos[k] < pos[p] for k, p in par.items())
            else:
                assert batch_size == 1
                count_done += 1
                seen.append(item.k)
            plan.done(item.k)
        assert count_done == n, batch_size
        assert sorted(seen) == sorted(make_tree().nodes())
        assert plan.completed()
    whole = make_tree().into_batched_workplan(n + 1).next().item
    assert whole.k == 1 and len(whole.subtree) == n


def test_wave_draining_of_a_batched_plan_has_no_repeats():
    """without done() in between, the reference yields the same subtree once per leaf anchor inside it"""
    plan = make_tree().into_batched_workplan(3)
    first = plan.next().item
    again = [plan.next().item for _ in range(3)]
    assert first.k == 43 and [a.k for a in again] == [9, 89, 9]  # 9 = {9, 0, 10} comes back for its second leaf
    plan2 = make_tree().into_batched_workplan(3)
    seen = []
    while True:
        wave = wp.drain_wave(plan2)
        if not wave:
            break
        assert len({it.k for it in wave}) == len(wave)
        for it in wave:
            seen += it.subtree.nodes()
            plan2.done(it.k)
    assert sorted(seen) == sorted(make_tree().nodes()) and plan2.completed()


def test_not_yet_until_done():
    plan = make_tree().into_workplan()
    wave = wp.drain_wave(plan)
    assert len(wave) == 4
    nxt = plan.next()
    assert nxt is not None and not nxt.ready  # Next::NotYet: parents wait for done()
    plan.done(1874)
    assert plan.next().item.k == 43
    assert not plan.completed()


def test_failures_mirror_reference_panics():
    with pytest.raises(mp2.Mp2gError):
        wp.UpdateTree.from_path([], 0)  # "empty path"
    with pytest.raises(mp2.Mp2gError):
        wp.UpdateTree.from_paths([[1, 2, 3], [1, 4, 2]], 0)  # duplicated key found in path
    mt = make_tree()
    with pytest.raises(mp2.Mp2gError):
        mt.extend_with_path([2, 3])  # assert_eq!(k, root)
    plan = mt.into_workplan()
    with pytest.raises(mp2.Mp2gError):
        plan.done(424242)  # RyhopeError::KeyNotFound
    with pytest.raises(mp2.Mp2gError):
        len(mt)  # consumed by the plan


def test_assign_subtrees_balances_and_is_deterministic():
    paths = [[0, a, 10 * a + b, 100 * a + 10 * b + c] for a in range(1, 5) for b in range(3) for c in range(1, 4)]
    plan = wp.UpdateTree.from_paths(paths, 0).into_batched_workplan(4)
    wave = wp.drain_wave(plan)
    owners = wp.assign_subtrees(wave, 4)
    assert owners == wp.assign_subtrees(wave, 4)
    loads = [sum(len(it.subtree) for it, o in zip(wave, owners) if o == r) for r in range(4)]
    assert max(loads) - min(loads) <= 4


def test_from_map():
    """updatetree.rs:296-331: pre-order arena, missing children skipped, is_path_end = the context is a leaf"""
    nodes = {8: (4, 12), 4: (2, 6), 12: (10, None), 2: (None, None), 6: (5, 7), 5: (None, None), 10: (None, 11), 99: (None, None)}
    t = wp.UpdateTree.from_map(5, 8, nodes)  # 7 and 11 are referenced but not in the map; 99 is unreachable
    assert t.epoch == 5 and t.nodes() == [8, 4, 2, 6, 5, 12, 10]
    par = t.parents()
    assert par == {8: None, 4: 8, 2: 4, 6: 4, 5: 6, 12: 8, 10: 12}
    assert t.path_ends() == {2, 5}  # 10 has a right key (absent from the map): not a NodeContext leaf
    plan = t.into_workplan()
    order = []
    while True:
        wave = wp.drain_wave(plan)
        if not wave:
            break
        for it in wave:
            order.append(it.k)
            plan.done(it.k)
    assert plan.completed() and sorted(order) == sorted(par) and order.index(5) < order.index(6) < order.index(4) < order.index(8)
    assert len(wp.UpdateTree.from_map(0, 1, {})) == 0
    with pytest.raises(mp2.Mp2gError):
        wp.UpdateTree.from_map(0, 1, {1: (2, 2), 2: (None, None)})  # duplicated key found


def test_forest_units_shrink_towards_the_end_of_a_wave(mp2):
    """mp2g_forest_group_units (pure host arithmetic, no GPU): the items of a wave stay whole and in order; units are capped at
    group_nodes and at an even share per worker; towards the end of the wave they shrink to half of what is left per worker and never
    below group_nodes / 6 -- so the last units are short and no worker idles for a whole unit"""
    import ctypes
    import numpy as np
    lib = mp2.load()

    def units(sizes, workers, group):
        a = np.asarray(sizes, dtype=np.uint32)
        first = np.zeros(len(a) + 1, dtype=np.uint32)
        n = ctypes.c_uint32()
        mp2._ck(lib.mp2g_forest_group_units(mp2._p(a), len(a), workers, group, mp2._p(first), ctypes.byref(n)))
        f = first[:n.value + 1].tolist()
        assert f[0] == 0 and f[-1] == len(a) and all(x < y for x, y in zip(f, f[1:]))
        return [int(a[lo:hi].sum()) for lo, hi in zip(f, f[1:])]

    # the driver's block: 512 subtrees of 40 rows, 4 workers, units of <= 1536 rows
    u = units([40] * 512, 4, 1536)
    assert sum(u) == 20480 and max(u) == 1560 and u[0] == 1560          # whole items: 39 x 40 rows pass the 1536 mark
    assert u[-1] <= 280 and min(u[:-1]) >= 256                          # the tail: a sixth of the cap (the last unit takes what is left)
    assert all(x >= y for x, y in zip(u, u[1:-1]))                      # never growing
    assert sum(1 for x in u if x < 1536) >= 6
    # a narrow wave: at least one unit per worker, an even share each
    assert units([63] * 16, 4, 1536) == [252] * 4
    # one item, no items
    assert units([64], 4, 1536) == [64] and units([], 4, 1536) == []
