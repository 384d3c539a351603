"""Update tree and work plan (children before parents), the reference's scheduler.

Host mirror of ryhope/src/storage/updatetree.rs over the C ABI (`mp2g_update_tree_*`,
`mp2g_update_plan_*` in include/mp2g.h; the logic is C++ in csrc/workplan.hip): same names, same
argument meaning, same failure points (the reference panics / returns RyhopeError::KeyNotFound where
these raise Mp2gError). Keys are u64.

    tree = UpdateTree.from_paths([[1, 3, 57, 9, 0], [1, 3, 89, 20]], epoch=3)
    plan = tree.into_batched_workplan(4)        # or into_workplan()
    for nxt in plan:                            # Next.ready / Next.not_yet
        if nxt.ready: prove(nxt.item); plan.done(nxt.item.k)

`assign_subtrees` deals the Ready items of one wave to ranks (one spun-off subtree = the unit one GPU
proves locally, returning only its root proof: SURVEY 8(e)).
"""
import ctypes

import numpy as np

from . import Mp2gError, _ck, load

PLAN_FINISHED, PLAN_READY, PLAN_NOT_YET = 0, 1, 2


def _lib():
    lib = load()
    if not getattr(lib, "_workplan_typed", False):
        lib.mp2g_update_tree_size.restype = ctypes.c_uint32
        lib.mp2g_update_tree_epoch.restype = ctypes.c_int64
        lib.mp2g_update_tree_free.argtypes = [ctypes.c_void_p]
        lib.mp2g_update_plan_free.argtypes = [ctypes.c_void_p]
        lib._workplan_typed = True
    return lib


class UpdateTree:
    """updatetree.rs:19-242. Node 0 of the arena is the root."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_paths(cls, paths, epoch=0):
        paths = [list(p) for p in paths]
        keys = np.ascontiguousarray([k for p in paths for k in p], dtype=np.uint64)
        lens = np.ascontiguousarray([len(p) for p in paths], dtype=np.uint32)
        h = ctypes.c_void_p()
        _ck(_lib().mp2g_update_tree_from_paths(keys.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
                                               len(paths), ctypes.c_int64(epoch), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_map(cls, epoch, root, nodes):
        """updatetree.rs:296-331. nodes: {key: (left or None, right or None)} (NodeContext)."""
        keys = np.ascontiguousarray(list(nodes), dtype=np.uint64)
        lr = [nodes[int(k)] for k in keys]
        left = np.ascontiguousarray([l if l is not None else 0 for l, _ in lr], dtype=np.uint64)
        right = np.ascontiguousarray([r if r is not None else 0 for _, r in lr], dtype=np.uint64)
        hl = np.ascontiguousarray([l is not None for l, _ in lr], dtype=np.uint8)
        hr = np.ascontiguousarray([r is not None for _, r in lr], dtype=np.uint8)
        h = ctypes.c_void_p()
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _ck(_lib().mp2g_update_tree_from_map(vp(keys), vp(left), vp(right), vp(hl), vp(hr), len(keys), ctypes.c_uint64(root),
                                             ctypes.c_int64(epoch), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_path(cls, path, epoch=0):
        if len(path) == 0:
            raise Mp2gError("empty path")
        return cls.from_paths([path], epoch)

    def _live(self):
        if self.h is None:
            raise Mp2gError("update tree was consumed by a work plan")
        return self.h

    def extend_with_path(self, path):
        p = np.ascontiguousarray(list(path), dtype=np.uint64)
        _ck(_lib().mp2g_update_tree_extend_with_path(self._live(), p.ctypes.data_as(ctypes.c_void_p), len(p)))

    def __len__(self):
        return int(_lib().mp2g_update_tree_size(self._live()))

    @property
    def epoch(self):
        return int(_lib().mp2g_update_tree_epoch(self._live()))

    def contains_key(self, k):
        return bool(_lib().mp2g_update_tree_contains_key(self._live(), ctypes.c_uint64(k)))

    def _dump(self):
        n = len(self)
        keys, parents, ends = np.empty(n, np.uint64), np.empty(n, np.int32), np.empty(n, np.uint8)
        _ck(_lib().mp2g_update_tree_nodes(self._live(), keys.ctypes.data_as(ctypes.c_void_p),
                                          parents.ctypes.data_as(ctypes.c_void_p), ends.ctypes.data_as(ctypes.c_void_p)))
        return keys, parents, ends

    def nodes(self):
        """keys in arena order (root first)"""
        return [int(k) for k in self._dump()[0]]

    def root(self):
        return self.nodes()[0]

    def parents(self):
        """{key: parent key or None}"""
        keys, parents, _ = self._dump()
        return {int(k): (None if p < 0 else int(keys[p])) for k, p in zip(keys, parents)}

    def path_ends(self):
        keys, _, ends = self._dump()
        return {int(k) for k, e in zip(keys, ends) if e}

    def subtree_size(self, k):
        out = ctypes.c_uint32()
        _ck(_lib().mp2g_update_tree_subtree_size(self._live(), ctypes.c_uint64(k), ctypes.byref(out)))
        return out.value

    def bottom_up(self):
        """keys ordered children-before-parents (what a GPU handed this subtree walks)"""
        keys, parents, _ = self._dump()
        depth = np.zeros(len(keys), dtype=np.int64)
        for i in range(1, len(keys)):
            depth[i] = depth[parents[i]] + 1  # arena order puts a parent before its children
        return [int(keys[i]) for i in np.argsort(-depth, kind="stable")]

    def into_workplan(self):
        return UpdatePlan(self, 1)

    def into_batched_workplan(self, subtree_size):
        return UpdatePlan(self, subtree_size)

    def free(self):
        if self.h is not None:
            _lib().mp2g_update_tree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class WorkplanItem:
    """updatetree.rs:372-409: Node {k, is_path_end} or Subtree {k, subtree}."""

    def __init__(self, k, is_path_end, subtree):
        self.k, self.is_path_end, self.subtree = k, is_path_end, subtree

    def as_subtree(self):
        assert self.subtree is not None
        return self.subtree

    def as_node(self):
        assert self.subtree is None
        return self.is_path_end


class Next:
    """updatetree.rs:362-369"""

    def __init__(self, item=None):
        self.item = item

    @property
    def ready(self):
        return self.item is not None


class UpdatePlan:
    """updatetree.rs:422-541; consumes the tree."""

    def __init__(self, tree, subtree_size):
        h = ctypes.c_void_p()
        _ck(_lib().mp2g_update_plan_create(tree._live(), subtree_size, ctypes.byref(h)))
        self.h, self.batch_size = h, subtree_size
        self._tree = UpdateTree(tree.h)  # view for tree(); owned by the plan
        tree.h = None

    def tree(self):
        return self._tree

    def next(self):
        """None when every node is done, else Next (ready or not-yet)."""
        k, end, sub = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_void_p()
        st = _lib().mp2g_update_plan_next(self.h, ctypes.byref(k), ctypes.byref(end), ctypes.byref(sub))
        if st < 0:
            raise Mp2gError(load().mp2g_last_error().decode())
        if st == PLAN_FINISHED:
            return None
        if st == PLAN_NOT_YET:
            return Next()
        return Next(WorkplanItem(k.value, bool(end.value), UpdateTree(sub) if sub.value else None))

    def __iter__(self):
        return self

    def __next__(self):
        n = self.next()
        if n is None:
            raise StopIteration
        return n

    def done(self, k):
        _ck(_lib().mp2g_update_plan_done(self.h, ctypes.c_uint64(k)))

    def completed(self):
        return bool(_lib().mp2g_update_plan_completed(self.h))

    def free(self):
        if self.h is not None:
            self._tree.h = None
            _lib().mp2g_update_plan_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def drain_wave(plan):
    """All items that are Ready right now (the reference harness's inner `while let Some(Next::Ready)`,
    mp2-v1/tests/common/celltree.rs:54-189). A batched plan that is drained without `done` in between
    hands out the same subtree once per leaf anchor it contains (the anchors of a spun-off subtree stay
    queued until its root is done: updatetree.rs:481-515); those repeats are dropped here."""
    out, seen = [], set()
    while True:
        n = plan.next()
        if n is None or not n.ready:
            return out
        if n.item.k in seen:
            if n.item.subtree is not None:
                n.item.subtree.free()
            continue
        seen.add(n.item.k)
        out.append(n.item)


def assign_subtrees(items, world):
    """Deal one wave of Ready items to ranks, largest first onto the least-loaded rank (work = number of
    nodes = number of framework proofs). Deterministic, so every rank computes the same table without
    talking. Returns [rank of items[i]]."""
    sizes = [len(it.subtree) if it.subtree is not None else 1 for it in items]
    load_, owner = [0] * world, [0] * len(items)
    for i in sorted(range(len(items)), key=lambda i: (-sizes[i], i)):
        r = min(range(world), key=lambda r: (load_[r], r))
        owner[i] = r
        load_[r] += sizes[i]
    return owner
