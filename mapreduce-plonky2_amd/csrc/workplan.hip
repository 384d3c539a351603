// Work plan over an update tree: children before parents, optionally in spun-off subtrees.
// Host-only code (no kernels): this is the scheduler that decides which unit of proving work goes
// to which GPU next. Behaviour follows ryhope/src/storage/updatetree.rs (UpdateTree :19-242,
// UpdatePlan :422-541) including its quirks: anchors are consumed LIFO, `done` of a subtree root
// detaches it from its parent only, stale anchors of already-spun-off leaves may be handed out again
// as single-node subtrees of size 1 only if their parent link still exists (it never does after
// `done`), and subtree_size 0 degenerates to single-node subtrees.
#include <algorithm>
#include <exception>
#include <memory>
#include <new>
#include <set>
#include <unordered_map>
#include <utility>
#include <vector>
#include "ctx.h"

using namespace mp2g;

// No exception crosses the C ABI: every entry point that allocates (plain `new`, std::set / unordered_map / vector growth) runs
// inside guarded(), which turns bad_alloc and anything else into the library's error code + mp2g_last_error().
template <class F> static int guarded(int on_error, F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    fail("out of memory");
  } catch (const std::exception& e) {
    fail("internal error: %s", e.what());
  } catch (...) {
    fail("internal error");
  }
  return on_error;
}

struct UtNode {
  int32_t parent;            // -1: none
  std::set<uint32_t> children;  // BTreeSet<usize>: ascending arena index
  uint64_t k;
  bool is_path_end;
};
struct mp2g_update_tree {
  int64_t epoch = 0;
  std::vector<UtNode> nodes;
  std::unordered_map<uint64_t, uint32_t> idx;

  uint32_t subtree_size_i(uint32_t i) const {
    uint32_t s = 1;
    for (uint32_t c : nodes[i].children) s += subtree_size_i(c);
    return s;
  }
  void descendants(uint32_t i, std::vector<uint32_t>& out) const {  // pre-order, children ascending
    out.push_back(i);
    for (uint32_t c : nodes[i].children) descendants(c, out);
  }
  // rec_from_path (:116-142), iteratively; `path` is what remains below node `cur`
  int descend(uint32_t cur, const uint64_t* path, uint32_t len) {
    for (uint32_t q = 0; q < len; q++) {
      const uint64_t k = path[q];
      int32_t child = -1;
      for (uint32_t c : nodes[cur].children)
        if (nodes[c].k == k) { child = (int32_t)c; break; }
      if (child < 0) {
        if (idx.count(k)) return fail("duplicated key found in path: %llu", (unsigned long long)k);
        const uint32_t ni = (uint32_t)nodes.size();
        idx.emplace(k, ni);
        nodes[cur].children.insert(ni);
        nodes[cur].is_path_end = false;
        nodes.push_back(UtNode{(int32_t)cur, {}, k, q + 1 == len});
        child = (int32_t)ni;
      }
      cur = (uint32_t)child;
    }
    return 0;
  }
  int extend(const uint64_t* path, uint32_t len) {
    if (len == 0) return 0;  // extend_with_path ignores an empty path (:147)
    if (nodes.empty()) {     // from_path (:95-114)
      nodes.push_back(UtNode{-1, {}, path[0], len == 1});
      idx.emplace(path[0], 0u);
    } else if (nodes[0].k != path[0]) {
      return fail("path does not start at the root of the update tree");
    }
    return descend(0, path + 1, len - 1);
  }
  std::unique_ptr<mp2g_update_tree> spin_off(uint32_t new_root) const {  // :184-228
    std::unique_ptr<mp2g_update_tree> t(new mp2g_update_tree);
    t->epoch = epoch;
    std::vector<uint32_t> d;
    descendants(new_root, d);
    for (uint32_t o : d) {
      t->idx.emplace(nodes[o].k, (uint32_t)t->nodes.size());
      t->nodes.push_back(UtNode{-1, {}, nodes[o].k, nodes[o].is_path_end});
    }
    for (uint32_t o : d) {
      const uint32_t n = t->idx[nodes[o].k];
      if (o != new_root && nodes[o].parent >= 0) t->nodes[n].parent = (int32_t)t->idx[nodes[nodes[o].parent].k];
      for (uint32_t c : nodes[o].children) t->nodes[n].children.insert(t->idx[nodes[c].k]);
    }
    return t;
  }
};
struct mp2g_update_plan {
  mp2g_update_tree* t;
  uint32_t batch_size;
  std::vector<uint64_t> anchors;
};

extern "C" {
int mp2g_update_tree_from_paths(const uint64_t* keys, const uint32_t* path_lens, uint32_t n_paths, int64_t epoch,
                                mp2g_update_tree** out) {
  if (!out || (n_paths && (!keys || !path_lens))) return fail("invalid argument: null pointer");
  return guarded(1, [&]() -> int {
    std::unique_ptr<mp2g_update_tree> t(new mp2g_update_tree);
    t->epoch = epoch;
    const uint64_t* p = keys;
    for (uint32_t i = 0; i < n_paths; i++) {
      if (i == 0 && path_lens[0] == 0) return fail("empty path");
      if (t->extend(p, path_lens[i])) return 1;
      p += path_lens[i];
    }
    *out = t.release();
    return 0;
  });
}
// UpdateTree::from_map (:296-331): pre-order walk from `root` over a map key -> (left, right); a child key that
// is not in the map is skipped; is_path_end = the node's context has no children at all (NodeContext::is_leaf)
int mp2g_update_tree_from_map(const uint64_t* keys, const uint64_t* left, const uint64_t* right, const uint8_t* has_left,
                              const uint8_t* has_right, uint32_t n, uint64_t root, int64_t epoch, mp2g_update_tree** out) {
  if (!out || (n && (!keys || !left || !right || !has_left || !has_right))) return fail("invalid argument: null pointer");
  return guarded(1, [&]() -> int {
    std::unordered_map<uint64_t, uint32_t> ctx;
    for (uint32_t i = 0; i < n; i++) ctx.emplace(keys[i], i);
    std::unique_ptr<mp2g_update_tree> t(new mp2g_update_tree);
    t->epoch = epoch;
    // explicit stack: (key, parent arena index); children pushed right first so that left is visited first
    std::vector<std::pair<uint64_t, int32_t>> stack{{root, -1}};
    while (!stack.empty()) {
      auto [k, parent] = stack.back();
      stack.pop_back();
      auto it = ctx.find(k);
      if (it == ctx.end()) continue;
      const uint32_t c = it->second, cur = (uint32_t)t->nodes.size();
      if (!t->idx.emplace(k, cur).second) return fail("duplicated key found");
      t->nodes.push_back(UtNode{parent, {}, k, !has_left[c] && !has_right[c]});
      if (parent >= 0) t->nodes[parent].children.insert(cur);
      if (has_right[c]) stack.push_back({right[c], (int32_t)cur});
      if (has_left[c]) stack.push_back({left[c], (int32_t)cur});
    }
    *out = t.release();
    return 0;
  });
}
int mp2g_update_tree_extend_with_path(mp2g_update_tree* t, const uint64_t* path, uint32_t len) {
  if (!t || (len && !path)) return fail("invalid argument: null pointer");
  if (t->nodes.empty() && len) return fail("extend_with_path on an empty update tree");
  // (a failure half way leaves the nodes inserted so far in place: they form a valid prefix of the path)
  return guarded(1, [&]() -> int { return t->extend(path, len); });
}
uint32_t mp2g_update_tree_size(const mp2g_update_tree* t) { return t ? (uint32_t)t->nodes.size() : 0; }
int64_t mp2g_update_tree_epoch(const mp2g_update_tree* t) { return t ? t->epoch : 0; }
int mp2g_update_tree_contains_key(const mp2g_update_tree* t, uint64_t k) { return t && t->idx.count(k) ? 1 : 0; }
int mp2g_update_tree_nodes(const mp2g_update_tree* t, uint64_t* keys, int32_t* parents, uint8_t* is_path_end) {
  if (!t) return fail("invalid argument: null tree");
  for (size_t i = 0; i < t->nodes.size(); i++) {
    if (keys) keys[i] = t->nodes[i].k;
    if (parents) parents[i] = t->nodes[i].parent;
    if (is_path_end) is_path_end[i] = t->nodes[i].is_path_end ? 1 : 0;
  }
  return 0;
}
int mp2g_update_tree_subtree_size(const mp2g_update_tree* t, uint64_t k, uint32_t* out) {
  if (!t || !out) return fail("invalid argument: null pointer");
  auto it = t->idx.find(k);
  if (it == t->idx.end() || it->second >= t->nodes.size()) return fail("key not found");
  *out = t->subtree_size_i(it->second);
  return 0;
}
void mp2g_update_tree_free(mp2g_update_tree* t) { delete t; }

int mp2g_update_plan_create(mp2g_update_tree* t, uint32_t subtree_size, mp2g_update_plan** out) {
  if (!t || !out) return fail("invalid argument: null pointer");
  return guarded(1, [&]() -> int {
    std::unique_ptr<mp2g_update_plan> p(new mp2g_update_plan);
    p->t = t;
    p->batch_size = subtree_size;
    for (const UtNode& n : t->nodes)  // every leaf is ready (:428-441)
      if (n.children.empty()) p->anchors.push_back(n.k);
    *out = p.release();
    return 0;
  });
}
int mp2g_update_plan_next(mp2g_update_plan* p, uint64_t* k, int* is_path_end, mp2g_update_tree** subtree) {
  if (!p || !k) { fail("invalid argument: null pointer"); return -1; }
  if (subtree) *subtree = nullptr;
  mp2g_update_tree& t = *p->t;
  if (t.nodes.empty()) return MP2G_PLAN_FINISHED;
  if (p->anchors.empty()) return MP2G_PLAN_NOT_YET;
  const uint64_t anchor = p->anchors.back();
  if (p->batch_size == 1) {
    auto it = t.idx.find(anchor);
    if (it == t.idx.end()) { fail("internal error: anchor not in the tree"); return -1; }
    p->anchors.pop_back();
    *k = anchor;
    if (is_path_end) *is_path_end = t.nodes[it->second].is_path_end ? 1 : 0;
    return MP2G_PLAN_READY;
  }
  if (!subtree) { fail("invalid argument: a batched plan needs the subtree out-pointer"); return -1; }
  // furthest ancestor whose subtree still fits the batch size (:481-515). The anchor leaves the list only once the subtree
  // exists: an allocation failure inside spin_off returns -1 with the plan unchanged.
  return guarded(-1, [&]() -> int {
    auto it = t.idx.find(anchor);
    if (it == t.idx.end()) { fail("internal error: anchor not in the tree"); return -1; }
    uint32_t root = it->second;
    while (t.nodes[root].parent >= 0) {
      const uint32_t parent = (uint32_t)t.nodes[root].parent;
      if (t.subtree_size_i(parent) > p->batch_size) break;
      root = parent;
    }
    std::unique_ptr<mp2g_update_tree> sub = t.spin_off(root);
    p->anchors.pop_back();
    *k = t.nodes[root].k;
    if (is_path_end) *is_path_end = t.nodes[root].is_path_end ? 1 : 0;
    *subtree = sub.release();
    return MP2G_PLAN_READY;
  });
}
int mp2g_update_plan_done(mp2g_update_plan* p, uint64_t k) {
  if (!p) return fail("invalid argument: null plan");
  mp2g_update_tree& t = *p->t;
  auto it = t.idx.find(k);
  if (it == t.idx.end()) return fail("key not found");
  const uint32_t i = it->second;
  return guarded(1, [&]() -> int {
    if (i != 0) {
      if (i >= t.nodes.size()) return fail("key not found");  // plan already finished
      const uint32_t parent = (uint32_t)t.nodes[i].parent;
      if (t.nodes[parent].children.size() == 1 && t.nodes[parent].children.count(i))
        p->anchors.reserve(p->anchors.size() + 1);  // the one step that can allocate comes first: a failure leaves the plan unchanged
    }
    p->anchors.erase(std::remove(p->anchors.begin(), p->anchors.end(), k), p->anchors.end());
    if (i == 0) {
      t.nodes.clear();
    } else {
      const uint32_t parent = (uint32_t)t.nodes[i].parent;
      t.nodes[parent].children.erase(i);
      if (t.nodes[parent].children.empty()) p->anchors.push_back(t.nodes[parent].k);
    }
    return 0;
  });
}
int mp2g_update_plan_completed(const mp2g_update_plan* p) { return p && p->t->nodes.empty() ? 1 : 0; }
void mp2g_update_plan_free(mp2g_update_plan* p) {
  if (!p) return;
  delete p->t;
  delete p;
}
}
