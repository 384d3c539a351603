// A forest of framework proofs proved bottom-up on the device: the native scheduler of a tree build (include/mp2g.h, mp2g_forest_*).
//
// Replaces the harness loops of the reference that prove every node of a tree children-before-parents, one
// RecursiveCircuits::generate_proof per node over its children's proofs (mp2-v1/tests/common/celltree.rs:54-189 for a row's cells
// tree, rowtree.rs:78-337 for the row tree; recursion-framework/src/framework.rs generate_proof) for a host that hands whole blocks of
// rows to a GPU. A node is registered with its circuit, its children and the words of its witness inputs that are not child proofs
// (circuit-set digest, the children's verifier data and membership proofs, the circuit's own inputs). Work arrives as UNITS -- lists of
// nodes, e.g. the spun-off subtrees of one wave of the update plan (workplan.hip) -- which W worker threads (one mp2g_ctx = HIP stream and
// one mp2g_chain per circuit each) take from a queue; a worker proves its unit level by level (levels counted inside the unit, so
// several subtrees' levels merge), each level's nodes of one circuit in batches of the chains' capacity. Final proofs live in a device
// pool (one fixed-size slot each: public inputs, the three proof caps, openings, FRI words -- the order a parent's witness inputs take
// them in); a parent's inputs receive its children's slots by one gather kernel, a batch's outputs reach their slots by another, and a
// slot is returned when the parent is proved. No proof visits the host unless it is asked for (mp2g_forest_proof).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "chain.h"

using namespace mp2g;
// the error text of a worker thread travels to the caller's thread (mp2g_last_error is thread local)
extern "C" const char* mp2g_last_error(void);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NEED(c, msg) do { if (!(c)) return fail("invalid argument: %s", msg); } while (0)

namespace {
constexpr uint32_t MAX_CHILDREN = 4;
struct Node {
  uint64_t id = 0;
  uint32_t circuit = 0;
  uint32_t n_children = 0;
  uint64_t child[MAX_CHILDREN] = {0, 0, 0, 0};
  size_t consts = 0;        // offset into the circuit's constant store
  int32_t slot = -1;        // pool slot of the proof, -1 = not proved (or released)
  uint32_t proof_words = 0; // words of the proof in its slot
  bool keep = false;        // the slot survives the parent (a checker downloads the proof later)
  bool proved = false;      // PUBLISHED: the device has written the slot and the witness check of its batch passed (set in confirm_oldest)
  uint32_t queued_by = 0;   // worker + 1 of the unit whose stream holds the node's batch, queued and not yet confirmed; 0 = none
};
struct Circuit {
  mp2g_forest_circuit d{};
  std::vector<u64> consts;  // [nodes of this circuit][n_const]
};
// copy jobs of the two gather kernels: words [src, src + n) -> [dst, dst + n)
struct Copy { const u64* src; u64* dst; uint32_t n; uint32_t pad; };
__global__ void __launch_bounds__(256) forest_copy_kernel(const Copy* jobs) {
  const Copy c = jobs[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < c.n; i += 256) c.dst[i] = c.src[i];
}
}  // namespace

struct mp2g_forest {
  uint32_t n_workers = 0, n_circuits = 0, slot_words = 0, pool_slots = 0;
  std::vector<mp2g_ctx*> ctxs;
  std::vector<mp2g_chain*> chains;  // [worker][circuit]
  std::vector<Circuit> circuits;
  std::unordered_map<uint64_t, uint32_t> index;
  std::vector<Node> nodes;
  DevBuf pool;
  std::vector<int32_t> free_slots;
  std::mutex mu;  // free_slots, node state that crosses workers (slot / proved of a child proved by another worker in an EARLIER call)
  std::condition_variable slots_back;  // a worker that found the pool empty waits here for another worker's parents to be proved
  uint32_t active = 0, waiting = 0;    // worker threads inside mp2g_forest_prove / of those, waiting for slots (under mu)
  std::atomic<uint64_t> proved{0};
  bool pipelined = true;  // MP2G_FOREST_SYNC=1: one batch at a time with a synchronisation behind each (the A/B switch)
  // per worker: a ring of batch records -- pinned staging of the copy jobs and of the witness-check flags, their device copy, the event
  // behind the batch's last launch. A worker queues batch k + 1 (and k + 2) while batch k runs: the stream never drains between the
  // batches of a unit (recursion-framework/src/circuit_builder.rs:286-311 semantics are untouched: the same proofs, checked one batch late)
  static constexpr uint32_t RING = 3;
  struct Slot { Copy* h_jobs = nullptr; Copy* d_jobs = nullptr; uint32_t* h_flags = nullptr; hipEvent_t done = nullptr; };
  struct Worker { Slot ring[RING]; uint32_t cap_jobs = 0; uint32_t seq = 0; };
  std::vector<Worker> workers;
  ~mp2g_forest() {
    for (auto& w : workers)
      for (auto& r : w.ring) {
        if (r.h_jobs) (void)hipHostFree(r.h_jobs);
        if (r.d_jobs) (void)hipFree(r.d_jobs);
        if (r.h_flags) (void)hipHostFree(r.h_flags);
        if (r.done) (void)hipEventDestroy(r.done);
      }
  }
};

namespace {
struct BatchCtx {  // what the chain hooks of one batch need
  const Copy* d_jobs; uint32_t n_between, n_after;  // jobs [0, n_between) patch the inputs, [n_between, n_between + n_after) store the outputs
};
int hook_copy(const Copy* d_jobs, uint32_t count, hipStream_t s) {
  if (!count) return 0;
  hipLaunchKernelGGL(forest_copy_kernel, dim3(count), dim3(256), 0, s, d_jobs);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail("forest copy kernel: %s", hipGetErrorString(e));
}
int hook_between(void* u, mp2g_chain*, hipStream_t s) { auto* b = (BatchCtx*)u; return hook_copy(b->d_jobs, b->n_between, s); }
int hook_after(void* u, mp2g_chain*, hipStream_t s) { auto* b = (BatchCtx*)u; return hook_copy(b->d_jobs + b->n_between, b->n_after, s); }

// a batch that has been queued on the worker's stream and not yet confirmed
struct InFlight {
  uint32_t ring = 0, B = 0, out_words = 0;
  mp2g_chain* ch = nullptr;
  std::vector<uint32_t> nodes;   // node indices, in batch order
  std::vector<int32_t> slots;    // their pool slots
};

// one unit on one worker; errors are returned as the library's code with mp2g_last_error() set by the failing call
int prove_unit_body(mp2g_forest* f, uint32_t w, const uint64_t* ids, uint32_t count, std::deque<InFlight>& flight);
// one unit; an allocation failure between a batch's enqueue and its confirmation must not leave its nodes queued and its slots
// taken: whatever is still in flight is rolled back before the error is returned
int prove_unit(mp2g_forest* f, uint32_t w, const uint64_t* ids, uint32_t count) {
  std::deque<InFlight> flight;  // queued batches, oldest first
  try {
    return prove_unit_body(f, w, ids, count, flight);
  } catch (...) {
    (void)hipStreamSynchronize(f->ctxs[w]->stream);
    {
      std::lock_guard<std::mutex> g(f->mu);
      for (InFlight& b : flight) {
        for (uint32_t i : b.nodes) { Node& n = f->nodes[i]; n.proved = false; n.queued_by = 0; n.slot = -1; n.proof_words = 0; }
        for (int32_t sl : b.slots) f->free_slots.push_back(sl);
      }
    }
    f->slots_back.notify_all();
    return fail("forest: out of memory (or another exception) inside a unit of worker %u: its queued batches were rolled back", w);
  }
}
int prove_unit_body(mp2g_forest* f, uint32_t w, const uint64_t* ids, uint32_t count, std::deque<InFlight>& flight) {
  if (!count) return 0;
  CK(hipSetDevice(f->ctxs[w]->device));
  std::vector<uint32_t> unit(count);
  std::unordered_map<uint32_t, uint32_t> level;  // node index -> level inside the unit
  for (uint32_t i = 0; i < count; i++) {
    auto it = f->index.find(ids[i]);
    if (it == f->index.end()) return fail("forest: unknown node %llu in a unit", (unsigned long long)ids[i]);
    unit[i] = it->second;
    if (!level.emplace(it->second, ~0u).second) return fail("forest: node %llu is listed twice in a unit", (unsigned long long)ids[i]);
    // a unit may be submitted again after a failure: its confirmed batches stay proved (their slots are valid), and a node that is
    // proved already -- or queued by another worker's unit of this call -- must not be proved twice
    if (f->nodes[it->second].proved || f->nodes[it->second].queued_by)
      return fail("forest: node %llu is already proved (a unit that failed half way is resubmitted WITHOUT the nodes its confirmed batches proved: mp2g_forest_proof succeeds on exactly those)", (unsigned long long)ids[i]);
  }
  // levels by repeated relaxation over the unit (children may come after their parents in `ids`)
  uint32_t max_level = 0;
  {
    std::vector<uint32_t> order(unit);
    bool changed = true;
    for (uint32_t i : unit) level[i] = 0;
    for (uint32_t guard = 0; changed && guard <= count; guard++) {
      changed = false;
      for (uint32_t i : unit) {
        const Node& n = f->nodes[i];
        uint32_t l = 0;
        for (uint32_t k = 0; k < n.n_children; k++) {
          auto c = f->index.find(n.child[k]);
          if (c == f->index.end()) return fail("forest: node %llu names an unknown child", (unsigned long long)n.id);
          auto lv = level.find(c->second);
          if (lv != level.end()) l = std::max(l, lv->second + 1);
        }
        if (l != level[i]) { level[i] = l; changed = true; }
        max_level = std::max(max_level, l);
      }
    }
    if (changed) return fail("forest: the unit's nodes form a cycle");
  }
  std::vector<std::vector<uint32_t>> by;  // [level * n_circuits + circuit] -> node indices
  by.resize((size_t)(max_level + 1) * f->n_circuits);
  for (uint32_t i : unit) by[(size_t)level[i] * f->n_circuits + f->nodes[i].circuit].push_back(i);
  mp2g_forest::Worker& W = f->workers[w];
  hipStream_t stream = f->ctxs[w]->stream;
  const uint32_t depth = f->pipelined ? mp2g_forest::RING - 1 : 0;  // batches that may stay queued behind the one being assembled

  // the oldest queued batch has run: its witness-check flags, then its children's slots go back (a slot returns when the parent is proved)
  auto confirm_oldest = [&]() -> int {
    InFlight& b = flight.front();
    hipError_t e = hipEventSynchronize(W.ring[b.ring].done);
    if (e != hipSuccess) return fail("forest: a batch of worker %u failed on the device: %s", w, hipGetErrorString(e));
    int rc = chain_flags_check(b.ch, b.B, W.ring[b.ring].h_flags);
    if (rc) return rc;
    {
      std::lock_guard<std::mutex> g(f->mu);
      for (uint32_t i : b.nodes) {
        Node& n = f->nodes[i];
        n.proved = true;  // published only now: the slot is written and checked (another worker's unit may name it as a child from here on)
        n.queued_by = 0;
        for (uint32_t k = 0; k < n.n_children; k++) {
          Node& cn = f->nodes[f->index[n.child[k]]];
          if (!cn.keep && cn.slot >= 0) { f->free_slots.push_back(cn.slot); cn.slot = -1; }
        }
      }
    }
    f->slots_back.notify_all();
    f->proved += b.B;
    flight.pop_front();
    return 0;
  };
  // a failure: nothing queued counts -- the stream is drained, the queued batches' nodes are unproved again and their slots free
  auto roll_back = [&]() {
    (void)hipStreamSynchronize(stream);
    std::lock_guard<std::mutex> g(f->mu);
    for (InFlight& b : flight) {
      for (uint32_t i : b.nodes) { Node& n = f->nodes[i]; n.proved = false; n.queued_by = 0; n.slot = -1; n.proof_words = 0; }
      for (int32_t sl : b.slots) f->free_slots.push_back(sl);
    }
    flight.clear();
    f->slots_back.notify_all();
  };
#define UNIT_FAIL(expr) do { int rc_ = (expr); if (rc_) { std::string msg_ = mp2g_last_error(); roll_back(); return fail("%s", msg_.c_str()); } } while (0)

  for (uint32_t lvl = 0; lvl <= max_level; lvl++) {
    for (uint32_t c = 0; c < f->n_circuits; c++) {
      const std::vector<uint32_t>& todo = by[(size_t)lvl * f->n_circuits + c];
      if (todo.empty()) continue;
      mp2g_chain* ch = f->chains[(size_t)w * f->n_circuits + c];
      if (!ch) UNIT_FAIL(fail("forest: worker %u has no chain for circuit %u", w, c));
      const Circuit& C = f->circuits[c];
      const mp2g_chain::Step& s0 = ch->steps[0];
      const mp2g_chain::Step& L = ch->steps[ch->n_steps - 1];
      const uint32_t n_pi = (uint32_t)(L.n_probe - 4), cw = (uint32_t)(3 * L.cap_words), ow = (uint32_t)(2 * L.n_open), pw = (uint32_t)L.proof_words;
      const uint32_t out_words = n_pi + cw + ow + pw;
      if (out_words > f->slot_words) UNIT_FAIL(fail("forest: a proof of circuit %u has %u words, the pool's slots %u", c, out_words, f->slot_words));
      for (size_t lo = 0; lo < todo.size(); lo += ch->cap) {
        const uint32_t B = (uint32_t)std::min<size_t>(ch->cap, todo.size() - lo);
        while (flight.size() > depth) UNIT_FAIL(confirm_oldest());
        // slots for the batch's proofs. An empty pool is not yet a failure: this worker's queued batches hold children whose slots
        // return once they are confirmed, and other workers return slots as their parents are proved -- only when every worker of
        // the call waits is the pool too small for the frontier
        std::vector<int32_t> slots(B);
        for (;;) {
          {
            std::unique_lock<std::mutex> g(f->mu);
            if (f->free_slots.size() >= B) {
              for (uint32_t j = 0; j < B; j++) { slots[j] = f->free_slots.back(); f->free_slots.pop_back(); }
              break;
            }
            if (flight.empty()) {
              f->waiting++;
              bool stuck = false;
              while (f->free_slots.size() < B) {
                if (f->waiting >= f->active) { stuck = true; break; }
                f->slots_back.wait_for(g, std::chrono::milliseconds(20));
              }
              f->waiting--;
              if (stuck) {
                g.unlock();
                f->slots_back.notify_all();
                UNIT_FAIL(fail("forest: the proof pool is exhausted (%u slots): release roots or create a larger pool", f->pool_slots));
              }
              continue;
            }
          }
          UNIT_FAIL(confirm_oldest());
        }
        const uint32_t ring = W.seq++ % mp2g_forest::RING;  // free: at most RING - 1 batches are queued
        mp2g_forest::Slot& R = W.ring[ring];
        uint32_t nb = 0;
        // from here on the batch owns its slots: every failure path gives them back
        auto give_back = [&]() { std::lock_guard<std::mutex> g(f->mu); for (int32_t sl : slots) f->free_slots.push_back(sl); };
        const uint32_t need_jobs = B * (C.d.n_children + 4);
        if (need_jobs > W.cap_jobs) { give_back(); UNIT_FAIL(fail("forest: internal: copy-job staging too small")); }
        // the chain's input buffers alternate: the one filled now is not the one whose upload may still be queued
        const uint32_t which = f->pipelined ? (ch->in_flip++ & 1) : 0;
        u64* h_in = nullptr;
        { int rc = chain_input_buffer(ch, which, &h_in); if (rc) { give_back(); UNIT_FAIL(rc); } }
        // inputs: the node's constant words around its children's ranges; the children come from their pool slots
        int arc = [&]() -> int {
        for (uint32_t j = 0; j < B; j++) {
          const Node& n = f->nodes[todo[lo + j]];
          u64* dst = h_in + (size_t)j * s0.n_in;
          const u64* src = C.consts.data() + n.consts;
          uint32_t at = 0;  // position in the inputs
          for (uint32_t k = 0; k <= C.d.n_children; k++) {
            const uint32_t end = k < C.d.n_children ? C.d.child_offset[k] : C.d.n_inputs;
            if (end < at) return fail("forest: circuit %u: child ranges out of order", c);
            memcpy(dst + at, src, (size_t)(end - at) * sizeof(u64));
            src += end - at;
            at = end;
            if (k < C.d.n_children) {
              Node* ch_node = nullptr;
              {
                std::lock_guard<std::mutex> g(f->mu);
                ch_node = &f->nodes[f->index[n.child[k]]];
                // a child is usable when it is published, or when THIS unit queued it on this worker's stream (stream order puts its
                // batch before this one); a node queued by another worker's unit is not: its slot may not be written yet
                if (!(ch_node->proved || ch_node->queued_by == w + 1) || ch_node->slot < 0) return fail("forest: child %llu of node %llu is not proved (or was released)", (unsigned long long)n.child[k], (unsigned long long)n.id);
              }
              if (at + ch_node->proof_words > C.d.n_inputs) return fail("forest: circuit %u: a child proof runs past the inputs", c);
              R.h_jobs[nb++] = Copy{f->pool.p + (size_t)ch_node->slot * f->slot_words, s0.in.p + (size_t)j * s0.n_in + at, ch_node->proof_words, 0};
              at += ch_node->proof_words;
            }
          }
          // the descriptor and the children's proofs must account for every input word and every constant word of the node
          if (at != C.d.n_inputs || src != C.consts.data() + n.consts + C.d.n_const)
            return fail("forest: circuit %u: the node's constant words and its children's proofs do not add up to the circuit's %u inputs", c, C.d.n_inputs);
        }
        return 0;
        }();
        if (arc) { give_back(); UNIT_FAIL(arc); }
        const uint32_t n_between = nb;
        // outputs -> slots, in a parent's input order: public inputs, caps of oracles 1..3, openings, FRI words
        for (uint32_t j = 0; j < B; j++) {
          u64* slot = f->pool.p + (size_t)slots[j] * f->slot_words;
          R.h_jobs[nb++] = Copy{L.probe.p + (size_t)j * L.n_probe + 4, slot, n_pi, 0};
          R.h_jobs[nb++] = Copy{L.caps.p + ((size_t)j * L.P.n_oracles + 1) * L.cap_words, slot + n_pi, cw, 0};
          R.h_jobs[nb++] = Copy{L.openings.p + (size_t)j * L.n_open * 2, slot + n_pi + cw, ow, 0};
          R.h_jobs[nb++] = Copy{L.proof.p + (size_t)j * L.proof_words, slot + n_pi + cw + ow, pw, 0};
        }
        {
          hipError_t e = hipMemcpyAsync(R.d_jobs, R.h_jobs, (size_t)nb * sizeof(Copy), hipMemcpyHostToDevice, stream);
          if (e != hipSuccess) { give_back(); UNIT_FAIL(fail("forest: copy jobs upload: %s", hipGetErrorString(e))); }
        }
        memset(R.h_flags, 0, (size_t)ch->n_steps * ch->cap * sizeof(uint32_t));
        BatchCtx bc{R.d_jobs, n_between, nb - n_between};
        ChainHooks hooks{&bc, hook_between, hook_after};
        int rc = chain_enqueue(ch, B, which, &hooks, R.h_flags);
        if (!rc) { hipError_t e = hipEventRecord(R.done, stream); if (e != hipSuccess) rc = fail("forest: event record: %s", hipGetErrorString(e)); }
        if (rc) { std::string msg = mp2g_last_error(); (void)hipStreamSynchronize(stream); give_back(); roll_back(); return fail("%s", msg.c_str()); }
        // queued: later levels of this unit (the same stream) may name these nodes as children from now on; other workers' units
        // see them once their batch is confirmed (confirm_oldest publishes)
        InFlight fl;
        fl.ring = ring; fl.B = B; fl.out_words = out_words; fl.ch = ch; fl.slots = slots;
        fl.nodes.assign(todo.begin() + lo, todo.begin() + lo + B);
        {
          std::lock_guard<std::mutex> g(f->mu);
          for (uint32_t j = 0; j < B; j++) {
            Node& n = f->nodes[todo[lo + j]];
            n.slot = slots[j]; n.proof_words = out_words; n.queued_by = w + 1;
          }
        }
        flight.push_back(std::move(fl));
      }
    }
  }
  while (!flight.empty()) UNIT_FAIL(confirm_oldest());
#undef UNIT_FAIL
  return 0;
}
}  // namespace


extern "C" {
int mp2g_forest_create(uint32_t n_workers, mp2g_ctx* const* ctxs, uint32_t n_circuits, const mp2g_forest_circuit* circuits,
                       mp2g_chain* const* chains, uint32_t slot_words, uint32_t pool_slots, mp2g_forest** out) {
  NEED(out && ctxs && circuits && chains && n_workers >= 1 && n_circuits >= 1 && slot_words >= 1 && pool_slots >= 1, "workers / circuits / chains / pool");
  try {
    std::unique_ptr<mp2g_forest> f(new mp2g_forest);
    f->n_workers = n_workers; f->n_circuits = n_circuits; f->slot_words = slot_words; f->pool_slots = pool_slots;
    f->ctxs.assign(ctxs, ctxs + n_workers);
    f->chains.assign(chains, chains + (size_t)n_workers * n_circuits);
    f->circuits.resize(n_circuits);
    uint32_t max_jobs = 0;
    for (uint32_t c = 0; c < n_circuits; c++) {
      const mp2g_forest_circuit& d = circuits[c];
      NEED(d.n_children <= MAX_CHILDREN && d.n_const <= d.n_inputs, "circuit descriptor");
      f->circuits[c].d = d;
      for (uint32_t w = 0; w < n_workers; w++) {
        mp2g_chain* ch = f->chains[(size_t)w * n_circuits + c];
        if (!ch) continue;
        NEED(ch->steps[0].n_in == d.n_inputs, "a chain's base circuit takes another number of inputs than its descriptor says");
        NEED(ch->ctx == ctxs[w], "a worker's chains live on the worker's context");
        max_jobs = std::max(max_jobs, ch->cap * (d.n_children + 4));
      }
    }
    for (uint32_t w = 0; w < n_workers; w++) NEED(ctxs[w] && ctxs[w]->device == ctxs[0]->device, "one forest = one GPU: every worker's context on the same device");
    CK(hipSetDevice(ctxs[0]->device));
    hipError_t e = f->pool.alloc((size_t)slot_words * pool_slots * sizeof(u64));
    if (e != hipSuccess) return fail("forest pool of %u x %u words: %s", pool_slots, slot_words, hipGetErrorString(e));
    f->free_slots.resize(pool_slots);
    for (uint32_t i = 0; i < pool_slots; i++) f->free_slots[i] = (int32_t)(pool_slots - 1 - i);
    f->workers.resize(n_workers);
    uint32_t max_flags = 1;
    for (mp2g_chain* ch : f->chains) if (ch) max_flags = std::max(max_flags, ch->n_steps * ch->cap);
    for (auto& w : f->workers) {
      w.cap_jobs = std::max(1u, max_jobs);
      for (auto& r : w.ring) {
        CK(hipHostMalloc((void**)&r.h_jobs, (size_t)w.cap_jobs * sizeof(Copy), hipHostMallocDefault));
        CK(hipMalloc((void**)&r.d_jobs, (size_t)w.cap_jobs * sizeof(Copy)));
        CK(hipHostMalloc((void**)&r.h_flags, (size_t)max_flags * sizeof(uint32_t), hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
      }
    }
    { const char* e = getenv("MP2G_FOREST_SYNC"); f->pipelined = !(e && atoi(e)); }
    *out = f.release();
    return 0;
  } catch (const std::bad_alloc&) { return fail("out of memory"); } catch (...) { return fail("internal error"); }
}

int mp2g_forest_add_nodes(mp2g_forest* f, uint32_t circuit, uint32_t count, const uint64_t* ids, const uint64_t* child_ids, const uint64_t* consts,
                          const uint8_t* keep) {
  NEED(f && circuit < f->n_circuits && (ids || !count), "forest / circuit / ids");
  Circuit& C = f->circuits[circuit];
  NEED((child_ids || !C.d.n_children || !count) && (consts || !C.d.n_const || !count), "children / constant words");
  try {
    {
      std::unordered_map<uint64_t, uint32_t> seen;
      seen.reserve(count);
      for (uint32_t i = 0; i < count; i++)
        if (f->index.count(ids[i]) || !seen.emplace(ids[i], i).second) return fail("forest: node %llu registered twice", (unsigned long long)ids[i]);
    }
    for (uint32_t i = 0; i < count; i++) {
      Node n;
      n.id = ids[i]; n.circuit = circuit; n.n_children = C.d.n_children;
      for (uint32_t k = 0; k < C.d.n_children; k++) n.child[k] = child_ids[(size_t)i * C.d.n_children + k];
      n.consts = C.consts.size();
      n.keep = keep && keep[i];
      C.consts.insert(C.consts.end(), consts + (size_t)i * C.d.n_const, consts + (size_t)(i + 1) * C.d.n_const);
      f->index.emplace(n.id, (uint32_t)f->nodes.size());
      f->nodes.push_back(n);
    }
    return 0;
  } catch (const std::bad_alloc&) { return fail("out of memory"); } catch (...) { return fail("internal error"); }
}

int mp2g_forest_prove(mp2g_forest* f, const uint64_t* unit_nodes, const uint32_t* unit_offsets, uint32_t n_units) {
  NEED(f && (n_units == 0 || (unit_nodes && unit_offsets)), "forest / units");
  if (!n_units) return 0;
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{0};
  std::mutex err_mu;
  std::string err;
  auto leave = [&]() { { std::lock_guard<std::mutex> g(f->mu); if (f->active) f->active--; } f->slots_back.notify_all(); };
  auto work = [&](uint32_t w) {
    struct Leave { decltype(leave)& fn; ~Leave() { fn(); } } on_exit{leave};  // a worker that is gone cannot return slots: the waiters must know
    try {
      for (;;) {
        if (failed.load()) return;
        const uint32_t u = next.fetch_add(1);
        if (u >= n_units) return;
        int rc = prove_unit(f, w, unit_nodes + unit_offsets[u], unit_offsets[u + 1] - unit_offsets[u]);
        if (rc) {
          std::lock_guard<std::mutex> g(err_mu);
          if (!failed.exchange(1)) err = mp2g_last_error();
          return;
        }
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(err_mu);
      if (!failed.exchange(1)) err = std::string("internal error: ") + e.what();
    } catch (...) {
      std::lock_guard<std::mutex> g(err_mu);
      if (!failed.exchange(1)) err = "internal error";
    }
  };
  try {
    const uint32_t n_threads = std::min(f->n_workers, n_units);
    { std::lock_guard<std::mutex> g(f->mu); f->active = n_threads; f->waiting = 0; }
    std::vector<std::thread> ts;
    for (uint32_t w = 1; w < n_threads; w++) ts.emplace_back(work, w);
    work(0);
    for (auto& t : ts) t.join();
  } catch (...) {
    return fail("forest: could not start the worker threads");
  }
  if (failed.load()) return fail("%s", err.c_str());
  return 0;
}

// Units of one wave: the wave's items (item i = item_sizes[i] plan nodes, kept whole) in order, in units of about group_nodes
// nodes, never fewer units than workers -- and SHRINKING towards the end of the wave (half of what is left per worker, down to a
// sixth of group_nodes): the workers pull units in order, and a wave ends when the last unit does; with units of one size the last
// ones run beside idle workers (a 2^16-row block: the last 9 % of the proofs at 0.73 of the rate, profiles/r05/block_2p16_progress.txt).
// MP2G_FOREST_FIXED_UNITS=1 keeps one size (the A/B switch). unit u = items [unit_first_item[u], unit_first_item[u + 1]).
int mp2g_forest_group_units(const uint32_t* item_sizes, uint32_t n_items, uint32_t n_workers, uint32_t group_nodes, uint32_t* unit_first_item,
                            uint32_t* n_units) {
  NEED((item_sizes || !n_items) && unit_first_item && n_units && n_workers >= 1, "items / outputs / workers");
  static const bool fixed_units = [] { const char* e = getenv("MP2G_FOREST_FIXED_UNITS"); return e && atoi(e); }();
  size_t total = 0;
  for (uint32_t i = 0; i < n_items; i++) total += item_sizes[i];
  const size_t cap = std::max<size_t>(1, std::min<size_t>(group_nodes ? group_nodes : 1, (total + n_workers - 1) / n_workers));
  const size_t floor_nodes = std::max<size_t>(1, (group_nodes ? group_nodes : 1) / 6);
  size_t assigned = 0, in_group = 0;
  auto target_now = [&]() -> size_t {
    if (fixed_units) return cap;
    return std::min(cap, std::max(floor_nodes, (total - assigned) / (2 * (size_t)n_workers)));
  };
  size_t target = target_now();
  uint32_t nu = 0;
  unit_first_item[0] = 0;
  for (uint32_t i = 0; i < n_items; i++) {
    in_group += item_sizes[i];
    if (in_group >= target) { unit_first_item[++nu] = i + 1; assigned += in_group; in_group = 0; target = target_now(); }
  }
  if (in_group) unit_first_item[++nu] = n_items;
  *n_units = nu;
  return 0;
}

// The reference harness's loop (mp2-v1/tests/common/rowtree.rs:78-337 with into_batched_workplan): drain every item that is Ready,
// prove it, mark it done, until the plan is finished -- with the items of a wave grouped into units of about group_nodes plan nodes
// (never fewer units than workers while the wave has the items). A plan node k stands for the forest node k and for its n_satellites
// satellite nodes ((j + 1) << satellite_shift) | k, j < n_satellites (a row and the cells-tree nodes of that row).
int mp2g_forest_prove_plan(mp2g_forest* f, mp2g_update_plan* plan, uint32_t group_nodes, uint32_t n_satellites, uint32_t satellite_shift,
                           uint32_t* waves, uint32_t* items_per_wave, uint32_t max_waves) {
  NEED(f && plan && satellite_shift < 64 && (!n_satellites || satellite_shift > 0), "forest / plan / satellites");
  try {
    uint32_t n_waves = 0;
    for (;;) {
      // one wave: every item that is Ready now (a batched plan drained without `done` in between hands the same subtree out once per
      // leaf anchor it holds: repeats are dropped, as workplan.drain_wave does)
      std::vector<std::vector<uint64_t>> items;
      std::vector<uint64_t> roots;
      for (;;) {
        uint64_t k = 0;
        int end = 0;
        mp2g_update_tree* sub = nullptr;
        const int st = mp2g_update_plan_next(plan, &k, &end, &sub);
        if (st < 0) return 1;
        if (st != MP2G_PLAN_READY) break;
        if (std::find(roots.begin(), roots.end(), k) != roots.end()) { if (sub) mp2g_update_tree_free(sub); continue; }
        std::vector<uint64_t> keys;
        if (sub) {
          keys.resize(mp2g_update_tree_size(sub));
          const int rc = mp2g_update_tree_nodes(sub, keys.data(), nullptr, nullptr);
          mp2g_update_tree_free(sub);
          if (rc) return rc;
        } else {
          keys.push_back(k);
        }
        roots.push_back(k);
        items.push_back(std::move(keys));
      }
      if (items.empty()) break;
      std::vector<uint32_t> sizes(items.size()), first(items.size() + 1);
      for (size_t i = 0; i < items.size(); i++) sizes[i] = (uint32_t)items[i].size();
      uint32_t n_units = 0;
      { const int rg = mp2g_forest_group_units(sizes.data(), (uint32_t)sizes.size(), f->n_workers, group_nodes, first.data(), &n_units); if (rg) return rg; }
      std::vector<uint64_t> nodes;
      std::vector<uint32_t> offs{0};
      for (uint32_t u = 0; u < n_units; u++) {
        for (uint32_t i = first[u]; i < first[u + 1]; i++)
          for (uint64_t k : items[i]) {
            for (uint32_t j = 0; j < n_satellites; j++) nodes.push_back(((uint64_t)(j + 1) << satellite_shift) | k);
            nodes.push_back(k);
          }
        offs.push_back((uint32_t)nodes.size());
      }
      const int rc = mp2g_forest_prove(f, nodes.data(), offs.data(), (uint32_t)offs.size() - 1);
      if (rc) return rc;
      for (uint64_t k : roots) {
        const int rd = mp2g_update_plan_done(plan, k);
        if (rd) return rd;
      }
      if (items_per_wave && n_waves < max_waves) items_per_wave[n_waves] = (uint32_t)items.size();
      n_waves++;
    }
    if (!mp2g_update_plan_completed(plan)) return fail("forest: the work plan stalled with items not done");
    if (waves) *waves = n_waves;
    return 0;
  } catch (const std::bad_alloc&) { return fail("out of memory"); } catch (...) { return fail("internal error"); }
}

int mp2g_forest_proof(mp2g_forest* f, uint64_t id, uint64_t* words, uint32_t* n_words) {
  NEED(f && n_words, "forest / outputs");
  const u64* src = nullptr;
  uint32_t n = 0;
  {
    std::lock_guard<std::mutex> g(f->mu);
    auto it = f->index.find(id);
    if (it == f->index.end()) return fail("forest: unknown node %llu", (unsigned long long)id);
    const Node& nd = f->nodes[it->second];
    if (!nd.proved || nd.slot < 0) return fail("forest: node %llu is not proved (or its proof was released)", (unsigned long long)id);
    src = f->pool.p + (size_t)nd.slot * f->slot_words;
    n = nd.proof_words;
  }
  *n_words = n;
  if (!words) return 0;
  CK(hipSetDevice(f->ctxs[0]->device));
  CK(hipMemcpy(words, src, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
  return 0;
}

int mp2g_forest_device_proof(mp2g_forest* f, uint64_t id, const uint64_t** d_words, uint32_t* n_words) {
  NEED(f && d_words && n_words, "forest / outputs");
  std::lock_guard<std::mutex> g(f->mu);
  auto it = f->index.find(id);
  if (it == f->index.end()) return fail("forest: unknown node %llu", (unsigned long long)id);
  const Node& nd = f->nodes[it->second];
  if (!nd.proved || nd.slot < 0) return fail("forest: node %llu is not proved (or its proof was released)", (unsigned long long)id);
  *d_words = f->pool.p + (size_t)nd.slot * f->slot_words;
  *n_words = nd.proof_words;
  return 0;
}

int mp2g_forest_release(mp2g_forest* f, uint64_t id) {
  NEED(f, "forest");
  std::lock_guard<std::mutex> g(f->mu);
  auto it = f->index.find(id);
  if (it == f->index.end()) return fail("forest: unknown node %llu", (unsigned long long)id);
  Node& nd = f->nodes[it->second];
  if (nd.slot >= 0) { f->free_slots.push_back(nd.slot); nd.slot = -1; }
  return 0;
}

uint64_t mp2g_forest_proved(const mp2g_forest* f) { return f ? f->proved.load() : 0; }
uint32_t mp2g_forest_free_slots(mp2g_forest* f) {
  if (!f) return 0;
  std::lock_guard<std::mutex> g(f->mu);
  return (uint32_t)f->free_slots.size();
}

void mp2g_forest_free(mp2g_forest* f) {
  if (!f) return;
  for (mp2g_ctx* c : f->ctxs) if (c) (void)hipStreamSynchronize(c->stream);
  delete f;
}
}  // extern "C"
