// codec.hip — whole-GOP entry points of the C-ABI: pcc_codec_create / pcc_encode_gop / pcc_decode_gop.
//
// One call = the whole of CompressionPipeline.compress() (sender/encoder/codec_pipeline.py:196-236)
// or DecompressionPipeline.decompress() (receiver/decoder/codec_parallel.py:141-171) for one GOP:
// stage order, coding order, container layout and every arithmetic step are those of the op-level
// entry points of pcc.h driven in the same sequence (the Python mirror in codec_pipeline.py /
// codec_parallel.py does exactly that, op by op), so both routes write byte-identical containers
// and reconstruct identical frames.  What this file adds is the host side in native code: the model
// graph (DESIGN.md MODEL), coordinate-set bookkeeping, a per-codec device pool instead of a tensor
// allocator, pinned staging, and the overlap of the serial host coders with the GPU (encoder: the
// stream runs analysis, hyper path and quantiser without a host synchronisation; the Q quality
// streams are then coded on Q threads while this thread produces the geometry slots — their kernels
// queue up behind the quantiser — and the z string; decoder: a helper thread decodes the z string
// while the coordinates are rebuilt, and the first synthesis rule book is built during the y decode).
//
// Host-only code (no kernels); compiled as HIP source for the runtime API.
#include "common.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <exception>
#include <new>
#include <map>
#include <memory>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int64_t kHashBuildMax = 60000;  // levels up to this many voxels hash directly; larger ones derive
// Decoder limits on what a container may ANNOUNCE before its streams have been decoded (a receiver is fed network bytes):
// latent rows per GOP (2^27 rows = a GOP of several 10^9 input points, beyond what 288 GB hold), and the number of z
// symbols for which the z job may start — and size its buffers — ahead of the geometry decode that verifies the counts.
constexpr int64_t kMaxLatentRows = (int64_t)1 << 27;
constexpr int64_t kEarlyZSymbols = (int64_t)1 << 22;

int log2i(int ts) {
  int s = 0;
  while ((1 << s) < ts) ++s;
  return s;
}

// ---------------------------------------------------------------- checkpoint blob
// "PCCW" | u32 count | count x { u16 name_len | name | u8 dtype (0 f32, 1 i32) | u8 ndim | u32 dims[ndim]
//                                | u64 nbytes | pad to 8 | data | pad to 8 }      (little-endian)
struct Tensor {
  int dtype = 0;
  std::vector<int64_t> dims;
  const uint8_t* data = nullptr;
  size_t nbytes = 0;
  int64_t numel() const {
    int64_t n = 1;
    for (int64_t d : dims) n *= d;
    return n;
  }
  const float* f32() const { return reinterpret_cast<const float*>(data); }
  const int32_t* i32() const { return reinterpret_cast<const int32_t*>(data); }
};

// ---------------------------------------------------------------- device pool
// Tensors of one call come from chained blocks that persist across calls: after the first GOP of a
// given size no call allocates device memory.
struct DevPool {
  struct Block {
    char* p;
    size_t cap, off;
  };
  std::vector<Block> blocks;
  // called with the stream idle.  A pool that has grown into many blocks (GOP sizes changing over time) is
  // re-made as one block of the same total size, so the chain stays short and nothing is stranded.
  void reset() {
    if (blocks.size() > 32) {
      size_t total = 0;
      for (auto& b : blocks) total += b.cap;
      release();
      char* p = nullptr;
      if (hipMalloc((void**)&p, total) == hipSuccess) blocks.push_back({p, total, 0});
    }
    for (auto& b : blocks) b.off = 0;
  }
  void* alloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    for (auto& b : blocks)
      if (b.cap - b.off >= bytes) {
        void* r = b.p + b.off;
        b.off += bytes;
        return r;
      }
    const size_t cap = std::max<size_t>(bytes, (size_t)64 << 20);
    char* p = nullptr;
    if (hipMalloc((void**)&p, cap) != hipSuccess) {
      pcc_set_error("codec pool: hipMalloc(%zu) failed", cap);
      return nullptr;
    }
    blocks.push_back({p, cap, bytes});
    return p;
  }
  void release() {
    for (auto& b : blocks) (void)hipFree(b.p);
    blocks.clear();
  }
};

// hipEvent that cannot leak on an early return
struct Event {
  hipEvent_t e = nullptr;
  int create() { return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? PCC_OK : PCC_E_HIP; }
  ~Event() {
    if (e) (void)hipEventDestroy(e);
  }
  Event() = default;
  Event(const Event&) = delete;
  Event& operator=(const Event&) = delete;
};

struct Pinned {
  uint8_t* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return PCC_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
    if (hipHostMalloc((void**)&p, want, hipHostMallocDefault) != hipSuccess) {
      pcc_set_error("codec: hipHostMalloc(%zu) failed", want);
      return PCC_E_NOMEM;
    }
    cap = want;
    return PCC_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// ---------------------------------------------------------------- coordinate sets
struct CS {
  uint64_t* keys = nullptr;  // device, Morton-sorted, unique
  int64_t n = 0;
  int stride = 1;
  int n_batch = 1;
  std::vector<int64_t> offsets;  // per-frame row offsets, filled on first use
  int32_t* nbr27 = nullptr;
  CS* down = nullptr;  // parents at stride*2 + the kernel-2 rule book [8, down->n]
  int32_t* nbr8 = nullptr;
  int32_t* parent_of = nullptr;
  CS* up = nullptr;          // generative children at stride/2, once made
  CS* gen_parent = nullptr;  // set whose generative children these rows are (row 8p+o)
  std::vector<int64_t> down_counts;  // row counts of the successive parent levels, when known (pcc_level_counts)
  CS* subset_of = nullptr;   // candidate set this set was pruned from, with the kept rows
  uint32_t* keep = nullptr;
  int32_t* keep_remap = nullptr;   // [subset_of->n] position of a candidate among the kept rows or -1, when the pruning wrote it
};

struct Feat {  // sparse tensor = coordinate set + feature rows
  CS* cs = nullptr;
  float* f = nullptr;
  int c = 0;
  // when set: row i of the tensor is row rows[i] of f (the survivors of a pruning, left in their candidates' tensor;
  // only up2 and the colour head read such a tensor, both through their *_gather entry points)
  const uint32_t* rows = nullptr;
};

}  // namespace

// The coder threads of a codec, kept alive between calls: starting three std::threads costs ~0.1 ms of a 3.4 ms
// encode before the last one runs; a parked thread takes its job within microseconds.  One job slot per worker.
struct PccWorkers {
  std::vector<std::thread> th;
  std::vector<std::function<void()>> job;
  std::vector<char> busy;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  bool stop = false;

  void ensure(int n) {
    std::unique_lock<std::mutex> lk(m);
    while ((int)th.size() < n) {
      const int i = (int)th.size();
      job.resize((size_t)i + 1);   // slots first: a thread that fails to start must not leave run(i) without a worker
      busy.resize((size_t)i + 1, 0);
      th.emplace_back([this, i]() {
        std::unique_lock<std::mutex> l(m);
        for (;;) {
          cv_job.wait(l, [&] { return stop || busy[i]; });
          if (stop) return;
          std::function<void()> f = std::move(job[i]);
          l.unlock();
          try {
            f();
          } catch (...) {  // jobs report through their own status words; never let an exception leave the thread
          }
          l.lock();
          busy[i] = 0;
          cv_done.notify_all();
        }
      });
    }
  }
  void run(int i, std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(m);
      job[i] = std::move(f);
      busy[i] = 1;
    }
    cv_job.notify_all();
  }
  void wait_range(int lo, int hi) {   // workers lo .. hi - 1 only (others may hold longer jobs)
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] {
      for (int i = lo; i < hi && i < (int)busy.size(); ++i)
        if (busy[i]) return false;
      return true;
    });
  }
  void wait_all() {
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] {
      for (char b : busy)
        if (b) return false;
      return true;
    });
  }
  ~PccWorkers() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_job.notify_all();
    for (auto& t : th)
      if (t.joinable()) t.join();
  }
};

struct pcc_codec {
  pcc_ctx* ctx = nullptr;
  int device = 0;
  std::vector<uint8_t> blob;
  std::map<std::string, Tensor> t;
  std::map<std::string, float*> dev;  // weights / biases / tables in HBM
  int c_y = 32, c_z = 32;
  PccRansTables* gc_tables = nullptr;  // coder tables of the Gaussian CDFs, built once (rans_gate.h)
  // container version written by pcc_encode_gop* (pcc_codec_set_container_version): 0 = the reference's layout, y and z
  // strings single rANS streams coded on the host; 1 = flagged extension, y and z strings wave-interleaved streams
  // coded on the GPU (rans_gpu.hip).  The decoder reads the version from the container.
  int container_version = 0;
  // seek points of the host-coded y strings (pcc_codec_set_seek_points): > 1 = version-0 containers get a "PCSK" trailer
  // behind their last frame record — the state and stream position of the y coder at that many cuts of the symbol array.
  // The reference's reader stops at the last frame record (codec_parallel.py:200-213) and never sees it; this library's
  // decoder decodes the pieces between the points on as many host threads.
  int seek_points = 0;
  // scale_nn(q) + eps rows on the device, kept while the same quality settings come in (every call of a service): two
  // slots, the encoder's Q rows and the decoder's one row.  (A per-call upload from a std::vector is a staged copy of
  // pageable memory: ~10 us of the calling thread and a launch on the stream for 128 bytes.)
  struct ScaleSlot {
    std::vector<double> key;
    float* dev = nullptr;
    size_t cap = 0;
  } scale_slot[2];
  pcc_rans_dev *gc_dev = nullptr, *eb_dev = nullptr;  // the two CDF sets in HBM for the GPU coder
  PccWorkers workers;                  // the Q coder threads of the encoder
  DevPool pool;
  Pinned pin_keys, pin_occ, pin_zsym, pin_ysym, pin_yidx, pin_flag, pin_dec, pin_up;
  hipStream_t up_stream = nullptr;     // host frames cross PCIe on this stream while the compute stream sorts (encode_gop_impl)
  // container version 1: the latent's octree kernel and the z string's coder run on a second stream (with a context
  // of its own: the operators take their scratch from their context's arena) beside h_a / h_s, whose launches leave
  // the chip all but empty (encode_gop_impl)
  hipStream_t side_stream = nullptr;
  pcc_ctx* side_ctx = nullptr;
  hipEvent_t up_done = nullptr;
  // events of a call (symbol pieces, geometry slot): created once, handed out again by every call (creating and
  // destroying half a dozen per GOP is host time on the path)
  std::vector<hipEvent_t> events;
  size_t events_used = 0;
  std::deque<CS> sets;
  std::vector<std::vector<uint8_t>> out;  // containers of the last encode
  // reconstruction of the last decode (device, pool-owned: valid until the next call)
  int32_t* rec_coords = nullptr;  // [n,4] (b,x,y,z)
  float* rec_colors = nullptr;    // [n,3]
  int64_t rec_n = 0;
  std::vector<int64_t> rec_offsets;
};

// ---------------------------------------------------------------- geometry slot (utils.py)
void octree_root(uint64_t first, uint64_t last, int key_shift, int* depth, int32_t origin[3]) {
  const uint64_t mask48 = ((uint64_t)1 << 48) - 1;
  const uint64_t a = (first & mask48) >> key_shift, b = (last & mask48) >> key_shift;
  const uint64_t diff = a ^ b;
  int d = 1;
  if (diff) d = (63 - __builtin_clzll(diff)) / 3 + 1;
  const uint64_t corner = (a >> (3 * d)) << (3 * d);
  const int bias = 32768 >> (key_shift / 3);
  auto compact = [](uint64_t v) {
    int r = 0;
    for (int i = 0; i < 16; ++i) r |= (int)((v >> (3 * i)) & 1) << i;
    return r;
  };
  *depth = d;
  origin[0] = compact(corner >> 2) - bias;
  origin[1] = compact(corner >> 1) - bias;
  origin[2] = compact(corner) - bias;
}


namespace {

#define CODEC_ALLOC(var, type, count)                                        \
  type* var = (type*)cd->pool.alloc(sizeof(type) * (size_t)(count));         \
  if (!var) return PCC_E_NOMEM

const Tensor* find(pcc_codec* cd, const std::string& name) {
  auto it = cd->t.find(name);
  if (it == cd->t.end()) {
    pcc_set_error("codec: checkpoint has no tensor '%s'", name.c_str());
    return nullptr;
  }
  return &it->second;
}

CS* new_set(pcc_codec* cd, uint64_t* keys, int64_t n, int stride, int n_batch) {
  cd->sets.emplace_back();
  CS* s = &cd->sets.back();
  s->keys = keys;
  s->n = n;
  s->stride = stride;
  s->n_batch = n_batch;
  return s;
}

int keys_of(pcc_codec* cd, CS* s, uint64_t** out);

int offsets_of(pcc_codec* cd, CS* s, const std::vector<int64_t>** out) {
  if (s->offsets.empty()) {
    if (s->n_batch == 1) {
      s->offsets = {0, s->n};
    } else if (s->gen_parent) {
      const std::vector<int64_t>* po;
      PCC_TRY(offsets_of(cd, s->gen_parent, &po));
      s->offsets.resize(po->size());
      for (size_t i = 0; i < po->size(); ++i) s->offsets[i] = 8 * (*po)[i];
    } else {
      s->offsets.assign((size_t)s->n_batch + 1, 0);
      PCC_TRY(pcc_batch_offsets(cd->ctx, s->keys, s->n, s->n_batch, s->offsets.data()));
    }
  }
  *out = &s->offsets;
  return PCC_OK;
}

int down_of(pcc_codec* cd, CS* s) {
  if (s->down) return PCC_OK;
  uint64_t* skeys;
  PCC_TRY(keys_of(cd, s, &skeys));
  const int64_t cap = std::max<int64_t>(s->n, 1);
  CODEC_ALLOC(pkeys, uint64_t, cap);
  CODEC_ALLOC(nbr8, int32_t, 8 * cap);
  CODEC_ALLOC(parent_of, int32_t, cap);
  int64_t m = 0;
  if (s->n > 0 && !s->down_counts.empty()) {  // size known: no read-back, the stream keeps running
    m = s->down_counts[0];
    PCC_TRY(pcc_down_coords_known(cd->ctx, skeys, s->n, 3 * log2i(s->stride), pkeys, nbr8, s->n, parent_of, m));
  } else if (s->n > 0) {
    PCC_TRY(pcc_down_coords(cd->ctx, skeys, s->n, 3 * log2i(s->stride), pkeys, nbr8, s->n, parent_of, &m));
  }
  s->down = new_set(cd, pkeys, m, s->stride * 2, s->n_batch);
  if (s->down_counts.size() > 1) s->down->down_counts.assign(s->down_counts.begin() + 1, s->down_counts.end());
  s->nbr8 = nbr8;
  s->parent_of = parent_of;
  return PCC_OK;
}

int up_of(pcc_codec* cd, CS* s, CS** out) {
  if (s->up) {
    *out = s->up;
    return PCC_OK;
  }
  if (s->stride < 2) {
    pcc_set_error("codec: cannot up-sample a stride-1 coordinate set");
    return PCC_E_ARG;
  }
  // no keys yet: row 8p + o is octant o of parent p, and what follows an up stage (the rule book derived from the
  // parents', the top-k's kept keys) works from the parents' keys — keys_of() writes the 8N keys for whoever asks
  CS* c = new_set(cd, nullptr, 8 * s->n, s->stride / 2, s->n_batch);
  c->gen_parent = s;
  s->up = c;
  *out = c;
  return PCC_OK;
}

int keys_of(pcc_codec* cd, CS* s, uint64_t** out) {
  if (!s->keys) {
    PCC_REQUIRE(s->gen_parent, PCC_E_ARG, "codec: coordinate set without keys");
    uint64_t* pk;
    PCC_TRY(keys_of(cd, s->gen_parent, &pk));
    CODEC_ALLOC(ckeys, uint64_t, std::max<int64_t>(s->n, 1));
    if (s->n > 0) PCC_TRY(pcc_up_coords(cd->ctx, pk, s->gen_parent->n, 3 * log2i(s->stride), ckeys));
    s->keys = ckeys;
  }
  *out = s->keys;
  return PCC_OK;
}

// 3^3 rule book of a set: derived from the parent level where there is one, hashed only at the
// coarsest level (sparse.py CoordSet._build_nbr27)
int nbr27_of(pcc_codec* cd, CS* s, int32_t** out) {
  if (!s->nbr27) {
    CODEC_ALLOC(nbr, int32_t, 27 * std::max<int64_t>(s->n, 1));
    CS* gp = s->gen_parent;
    if (s->n == 0) {
      // nothing to fill
    } else if (gp && gp->n > 0) {
      int32_t* pn;
      PCC_TRY(nbr27_of(cd, gp, &pn));
      PCC_TRY(pcc_derive_map_up(cd->ctx, pn, gp->n, nullptr, nullptr, gp->n, nbr));
    } else if (s->subset_of && s->subset_of->gen_parent && s->subset_of->gen_parent->n > 0) {
      // pruned level: straight from the book of the level its candidates were generated from
      CS* cand = s->subset_of;
      int32_t* pn;
      PCC_TRY(nbr27_of(cd, cand->gen_parent, &pn));
      int32_t* remap = s->keep_remap;
      if (!remap) {
        remap = (int32_t*)cd->pool.alloc(sizeof(int32_t) * (size_t)cand->n);
        if (!remap) return PCC_E_NOMEM;
        PCC_TRY(pcc_inverse_rows(cd->ctx, s->keep, s->n, cand->n, remap));
      }
      PCC_TRY(pcc_subset_map_up(cd->ctx, pn, cand->gen_parent->n, s->keep, remap, s->n, nbr));
    } else if (s->n > kHashBuildMax && s->stride <= 4096) {
      PCC_TRY(down_of(cd, s));
      int32_t* pn;
      PCC_TRY(nbr27_of(cd, s->down, &pn));
      uint64_t* skeys;
      PCC_TRY(keys_of(cd, s, &skeys));
      PCC_TRY(pcc_derive_map_down(cd->ctx, pn, s->down->n, s->nbr8, s->parent_of, skeys, s->n,
                                  3 * log2i(s->stride), nbr));
    } else {
      uint64_t* skeys;
      PCC_TRY(keys_of(cd, s, &skeys));
      PCC_TRY(pcc_build_map(cd->ctx, skeys, s->n, s->stride, nbr));
    }
    s->nbr27 = nbr;
  }
  *out = s->nbr27;
  return PCC_OK;
}

// ---------------------------------------------------------------- layers (model.py)
int wb(pcc_codec* cd, const std::string& name, const float** w, const float** b, const Tensor** wt) {
  const Tensor* tw = find(cd, name + ".weight");
  if (!tw) return PCC_E_ARG;
  *w = cd->dev[name + ".weight"];
  *b = cd->dev[name + ".bias"];
  *wt = tw;
  return PCC_OK;
}

int conv3(pcc_codec* cd, const std::string& name, const Feat& x, int relu, Feat* y) {
  const float *w, *b;
  const Tensor* tw;
  PCC_TRY(wb(cd, name, &w, &b, &tw));
  const int cin = (int)tw->dims[1], cout = (int)tw->dims[2];
  int32_t* nbr;
  PCC_TRY(nbr27_of(cd, x.cs, &nbr));
  CODEC_ALLOC(o, float, std::max<int64_t>(x.cs->n, 1) * cout);
  PCC_TRY(pcc_sparse_conv(cd->ctx, x.f, x.cs->n, nbr, 27, x.cs->n, x.cs->n, w, b, cin, cout, relu, o));
  *y = {x.cs, o, cout};
  return PCC_OK;
}

int down2(pcc_codec* cd, const std::string& name, const Feat& x, int relu, Feat* y) {
  const float *w, *b;
  const Tensor* tw;
  PCC_TRY(wb(cd, name, &w, &b, &tw));
  const int cin = (int)tw->dims[1], cout = (int)tw->dims[2];
  PCC_TRY(down_of(cd, x.cs));
  CS* p = x.cs->down;
  CODEC_ALLOC(o, float, std::max<int64_t>(p->n, 1) * cout);
  PCC_TRY(pcc_sparse_conv(cd->ctx, x.f, x.cs->n, x.cs->nbr8, 8, p->n, p->n, w, b, cin, cout, relu, o));
  *y = {p, o, cout};
  return PCC_OK;
}

// perm: the output rows are written in the channel order of pcc_conv16_perm (weight columns and bias permuted at
// pcc_codec_create) — for the up stages of g_s, whose output only pcc_sparse_conv_head_up_perm reads
int up2(pcc_codec* cd, const std::string& name, const Feat& x, int relu, Feat* y, bool perm = false) {
  const float *w, *b;
  const Tensor* tw;
  PCC_TRY(wb(cd, name, &w, &b, &tw));
  if (perm) {
    auto iw = cd->dev.find(name + ".weight#perm"), ib = cd->dev.find(name + ".bias#perm");
    PCC_REQUIRE(iw != cd->dev.end() && ib != cd->dev.end(), PCC_E_ARG, "codec: no permuted weights for '%s'", name.c_str());
    w = iw->second;
    b = ib->second;
  }
  const int cin = (int)tw->dims[1], cout = (int)tw->dims[2];
  CS* c;
  PCC_TRY(up_of(cd, x.cs, &c));
  CODEC_ALLOC(o, float, std::max<int64_t>(c->n, 1) * cout);
  if (x.rows) {
    PCC_REQUIRE(cin == 32 && cout == 32, PCC_E_ARG, "codec: row-indirect up stage needs 32 -> 32 channels");
    PCC_TRY(pcc_convT_gen_gather(cd->ctx, x.f, x.rows, x.cs->n, w, b, relu, o));
  } else {
    PCC_TRY(pcc_convT_gen(cd->ctx, x.f, x.cs->n, w, b, cin, cout, relu, o));
  }
  *y = {c, o, cout};
  return PCC_OK;
}

// canonical-order view of a set (utils.sort_coordset / sort_tensor): coordinates [n,4] in the
// reference's sort order and the permutation canonical position -> row
struct View {
  int32_t* coords = nullptr;
  uint32_t* perm = nullptr;
  int64_t n = 0;
};

int side_of(pcc_codec* cd, pcc_ctx** out) {
  if (!cd->side_ctx) {
    if (!cd->side_stream) PCC_HIP(hipStreamCreateWithFlags(&cd->side_stream, hipStreamNonBlocking));
    cd->side_ctx = pcc_create(cd->device, cd->side_stream);
    if (!cd->side_ctx) return PCC_E_HIP;
  }
  *out = cd->side_ctx;
  return PCC_OK;
}

// an event of the codec's pool for this call (cd->events_used is reset where the call resets the device pool)
int call_event(pcc_codec* cd, hipEvent_t* out) {
  if (cd->events_used == cd->events.size()) {
    hipEvent_t e = nullptr;
    PCC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    cd->events.push_back(e);
  }
  *out = cd->events[cd->events_used++];
  return PCC_OK;
}

int view_of(pcc_codec* cd, CS* s, View* v) {
  const int64_t cap = std::max<int64_t>(s->n, 1);
  CODEC_ALLOC(perm, uint32_t, cap);
  CODEC_ALLOC(cs, int32_t, 4 * cap);
  uint64_t* skeys;
  PCC_TRY(keys_of(cd, s, &skeys));
  if (s->n > 0 && s->n <= pcc_sort_small_max()) {
    // latent-sized set: order and ordered rows straight from the keys
    PCC_TRY(pcc_sort_keys_canonical(cd->ctx, skeys, s->n, perm, cs));
    *v = {cs, perm, s->n};
    return PCC_OK;
  }
  CODEC_ALLOC(c, int32_t, 4 * cap);
  if (s->n > 0) {
    PCC_TRY(pcc_keys_to_coords(cd->ctx, skeys, s->n, c));
    PCC_TRY(pcc_sort_coords(cd->ctx, c, s->n, perm));
    PCC_TRY(pcc_gather_rows(cd->ctx, c, perm, s->n, 16, cs));
  }
  *v = {cs, perm, s->n};
  return PCC_OK;
}

// rows given in canonical order -> rows of the (Morton-ordered) tensor (utils.sparse_from_rows)
int rows_to_tensor(pcc_codec* cd, const View& v, const float* rows, int c, float** out) {
  CODEC_ALLOC(o, float, std::max<int64_t>(v.n, 1) * c);
  if (v.n > 0) {
    if ((4 * c) % 16 == 0 && (uintptr_t)rows % 16 == 0) {   // v.perm is a permutation: one scatter
      PCC_TRY(pcc_scatter_rows(cd->ctx, rows, v.perm, v.n, 4 * c, o));
    } else {
      CODEC_ALLOC(inv, int32_t, v.n);
      PCC_TRY(pcc_inverse_rows(cd->ctx, v.perm, v.n, v.n, inv));
      PCC_TRY(pcc_gather_rows(cd->ctx, rows, (const uint32_t*)inv, v.n, 4 * c, o));
    }
  }
  *out = o;
  return PCC_OK;
}

// scale_nn(q) + eps on the host in float32 with the operation order of model.py ScaleNN
int scale_row(pcc_codec* cd, double qg, double qa, float* out /*[c_y]*/) {
  const Tensor *w0 = find(cd, "scale_nn.l0.weight"), *b0 = find(cd, "scale_nn.l0.bias");
  const Tensor *w1 = find(cd, "scale_nn.l1.weight"), *b1 = find(cd, "scale_nn.l1.bias");
  const Tensor* eps = find(cd, "entropy_model.eps");
  if (!w0 || !b0 || !w1 || !b1 || !eps) return PCC_E_ARG;
  const int hid = (int)w0->dims[1], co = (int)w1->dims[1];
  const float q[2] = {(float)qg, (float)qa};
  std::vector<float> h(b0->f32(), b0->f32() + hid);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < hid; ++j) {
      const float prod = q[i] * w0->f32()[i * hid + j];
      h[j] = h[j] + prod;
    }
  for (int j = 0; j < hid; ++j) h[j] = std::max(h[j], 0.0f);
  std::vector<float> o(b1->f32(), b1->f32() + co);
  for (int i = 0; i < hid; ++i)
    for (int j = 0; j < co; ++j) {
      const float prod = h[i] * w1->f32()[i * co + j];
      o[j] = o[j] + prod;
    }
  const float e = eps->f32()[0];
  for (int j = 0; j < co; ++j) {
    const float s = 0.5f + fabsf(o[j]);
    out[j] = s + e;
  }
  return PCC_OK;
}

// the rows scale_nn(q) + eps of n_q quality settings in HBM (cached per codec: see pcc_codec::scale_slot)
int scale_rows_dev(pcc_codec* cd, int slot, const double* h_q, int n_q, float** out) {
  const int cy = cd->c_y;
  pcc_codec::ScaleSlot& sl = cd->scale_slot[slot];
  const std::vector<double> key(h_q, h_q + 2 * n_q);
  if (sl.dev && sl.key == key) {
    *out = sl.dev;
    return PCC_OK;
  }
  std::vector<float> h((size_t)n_q * cy);
  for (int q = 0; q < n_q; ++q) PCC_TRY(scale_row(cd, h_q[2 * q], h_q[2 * q + 1], &h[(size_t)q * cy]));
  if (sl.cap < h.size() * 4) {
    PCC_HIP(hipStreamSynchronize(cd->ctx->stream));
    if (sl.dev) PCC_HIP(hipFree(sl.dev));
    sl.dev = nullptr;
    sl.cap = 0;
    PCC_HIP(hipMalloc((void**)&sl.dev, h.size() * 4));
    sl.cap = h.size() * 4;
  }
  sl.key.clear();   // valid again only once the copy below has been queued
  PCC_HIP(hipMemcpyAsync(sl.dev, h.data(), h.size() * 4, hipMemcpyHostToDevice, cd->ctx->stream));
  PCC_HIP(hipStreamSynchronize(cd->ctx->stream));   // h leaves scope; once per change of settings
  sl.key = key;
  *out = sl.dev;
  return PCC_OK;
}

// h_s up to its output layer: the 64 generated descendants of every z voxel at stride 8 with their features
int h_s_up(pcc_codec* cd, const Feat& z_hat, Feat* pre) {
  Feat a;
  PCC_TRY(up2(cd, "h_s.up0", z_hat, 1, &a));
  return up2(cd, "h_s.up1", a, 1, pre);
}

// h_s output layer + features_at_coordinates(qcoords) in one: the reference evaluates the 32 -> 64 conv on every
// descendant (4x the latent's rows) and then samples it at the latent's coordinates; a row's value depends only on its
// own neighbour list, so the conv is run on the sampled rows alone (their rule-book columns), absent rows -> 0.
int h_s_out_at(pcc_codec* cd, const Feat& pre, const CS* ycs, const View& yv, float** out) {
  const int32_t* qcoords = yv.coords;
  const int64_t m = yv.n;
  const float *w, *b;
  const Tensor* tw;
  PCC_TRY(wb(cd, "h_s.conv0", &w, &b, &tw));
  const int cin = (int)tw->dims[1], cout = (int)tw->dims[2];
  const int64_t cap = std::max<int64_t>(m, 1);
  CODEC_ALLOC(nbr_sub, int32_t, 27 * cap);
  CODEC_ALLOC(o, float, cap * cout);
  if (m > 0) {
    const CS* z16 = ycs->down;
    const CS* z32 = z16 ? z16->down : nullptr;
    const bool by_structure = ycs->stride == 8 && ycs->n == m && z32 && ycs->parent_of && z16->parent_of &&
                              pre.cs->gen_parent && pre.cs->gen_parent->gen_parent == z32;
    if (by_structure) {
      // pre = up(up(z)) and z = down(down(y)): every latent voxel IS one of the 64 descendants of its stride-32
      // ancestor, its row and its rule-book column follow from the two parent maps and the book one level up — no
      // hash table, no lookup, no book of the descendants, and no row is absent (o = the conv's output as it is)
      int32_t* pn;
      PCC_TRY(nbr27_of(cd, pre.cs->gen_parent, &pn));
      PCC_TRY(pcc_descendant_map(cd->ctx, pn, pre.cs->gen_parent->n, (const uint32_t*)yv.perm, (const uint64_t*)ycs->keys,
                                 (const int32_t*)ycs->parent_of, (const int32_t*)z16->parent_of, m, nbr_sub));
      PCC_TRY(pcc_sparse_conv(cd->ctx, pre.f, pre.cs->n, nbr_sub, 27, m, m, w, b, cin, cout, 0, o));
    } else {  // general form: hash the descendants' keys and look the coordinates up
      int32_t* nbr;
      PCC_TRY(nbr27_of(cd, pre.cs, &nbr));
      CODEC_ALLOC(qkeys, uint64_t, cap);
      CODEC_ALLOC(flag, int32_t, 1);
      CODEC_ALLOC(rows, int32_t, cap);
      CODEC_ALLOC(self, int32_t, cap);
      CODEC_ALLOC(conv_o, float, cap * cout);
      PCC_HIP(hipMemsetAsync(flag, 0, 4, cd->ctx->stream));
      PCC_TRY(pcc_morton_keys(cd->ctx, qcoords, m, qkeys, flag));
      uint64_t* pre_keys;
      PCC_TRY(keys_of(cd, pre.cs, &pre_keys));
      PCC_TRY(pcc_lookup(cd->ctx, pre_keys, pre.cs->n, qkeys, m, rows));
      PCC_TRY(pcc_gather_map_columns(cd->ctx, nbr, 27, pre.cs->n, rows, m, nbr_sub, self));
      PCC_TRY(pcc_sparse_conv(cd->ctx, pre.f, pre.cs->n, nbr_sub, 27, m, m, w, b, cin, cout, 0, conv_o));
      PCC_TRY(pcc_gather_rows_or_zero(cd->ctx, conv_o, self, m, cout, o));
    }
  }
  *out = o;
  return PCC_OK;
}

void put_be32(std::vector<uint8_t>& v, int32_t x) {
  const uint32_t u = (uint32_t)x;
  v.push_back((uint8_t)(u >> 24));
  v.push_back((uint8_t)(u >> 16));
  v.push_back((uint8_t)(u >> 8));
  v.push_back((uint8_t)u);
}
void put_be_f64(std::vector<uint8_t>& v, double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  for (int i = 7; i >= 0; --i) v.push_back((uint8_t)(u >> (8 * i)));
}

struct Reader {
  const uint8_t* p;
  int64_t len, pos = 0;
  bool bad = false;
  int32_t be32() {
    if (pos + 4 > len) { bad = true; return 0; }
    const uint32_t u = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
    pos += 4;
    return (int32_t)u;
  }
  double be_f64() {
    if (pos + 8 > len) { bad = true; return 0; }
    uint64_t u = 0;
    for (int i = 0; i < 8; ++i) u = (u << 8) | p[pos + i];
    pos += 8;
    double d;
    memcpy(&d, &u, 8);
    return d;
  }
  const uint8_t* bytes(int64_t n) {
    if (n < 0 || pos + n > len) { bad = true; return nullptr; }
    const uint8_t* r = p + pos;
    pos += n;
    return r;
  }
};

int rans_encode_grow(const int32_t* sym, const int32_t* idx, int64_t n, const Tensor* cdf, const Tensor* len,
                     const Tensor* off, std::vector<uint8_t>* out) {
  int64_t cap = 2 * n + 4096, got = 0;
  int rc = PCC_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    out->resize((size_t)cap);
    rc = pcc_rans_encode(sym, idx, n, cdf->i32(), (int)cdf->dims[1], len->i32(), off->i32(), (int)cdf->dims[0],
                         out->data(), cap, &got);
    if (rc != PCC_E_NOMEM) break;
    cap = 48 * n + 4096;
  }
  if (rc == PCC_OK) out->resize((size_t)got);
  return rc;
}

}  // namespace

// ======================================================================== C-ABI

// The runtime sets a transfer path up the first time it is used (another copy engine while the first is busy, a size
// class, a direction): ~5 ms, once per process — and when that first time fell into a GOP, that GOP took twice as long
// (seen as one run in four of bench.py reading 96 instead of 100 frames/s).  A codec therefore walks through transfers
// of the kinds its calls make — small and medium ones, both directions, several in flight — when it is created.
// Best effort: a failure here is not an error of the codec.
static void warm_copy_paths(pcc_codec* cd) {
  hipStream_t st = cd->ctx->stream;
  const size_t big = (size_t)1 << 20;
  uint8_t *d = nullptr, *h = nullptr;
  if (hipMalloc((void**)&d, 2 * big) != hipSuccess) return;
  if (hipHostMalloc((void**)&h, 2 * big, hipHostMallocDefault) == hipSuccess) {
    for (int rep = 0; rep < 2; ++rep) {
      for (size_t bytes : {(size_t)64, (size_t)4096, (size_t)65536, big}) {
        for (int k = 0; k < 4; ++k) {
          (void)hipMemcpyAsync(h + (size_t)k * (big / 4), d + (size_t)k * (big / 4), std::min(bytes, big / 4), hipMemcpyDeviceToHost, st);
          (void)hipMemcpyAsync(d + big + (size_t)k * (big / 4), h + big + (size_t)k * (big / 4), std::min(bytes, big / 4),
                               hipMemcpyHostToDevice, st);
        }
        // (the symbol pieces of a version-0 encode: a few rows of up to several hundred KB each)
        (void)hipMemcpy2DAsync(h, big / 4, d, big / 4, std::min(bytes, big / 4), 3, hipMemcpyDeviceToHost, st);
        (void)hipMemcpy2DAsync(h + big, big / 4, d + big, big / 4, std::min(bytes, big / 4) / 2 + 1, 3, hipMemcpyDeviceToHost, st);
      }
      (void)hipMemsetAsync(d, 0, 4096, st);
      (void)hipMemcpyAsync(d + big, d, 65536, hipMemcpyDeviceToDevice, st);
    }
    (void)hipStreamSynchronize(st);
    (void)hipHostFree(h);
  }
  (void)hipFree(d);
  (void)hipGetLastError();
}

extern "C" pcc_codec* pcc_codec_create(const void* h_ckpt, size_t n, int device, void* stream) {
  if (!h_ckpt || n < 8 || memcmp(h_ckpt, "PCCW", 4) != 0) {
    pcc_set_error("pcc_codec_create: not a PCCW checkpoint blob");
    return nullptr;
  }
  pcc_ctx* ctx = pcc_create(device, stream);
  if (!ctx) return nullptr;
  pcc_codec* cd = new pcc_codec();
  cd->ctx = ctx;
  cd->device = device;
  cd->blob.assign((const uint8_t*)h_ckpt, (const uint8_t*)h_ckpt + n);
  const uint8_t* p = cd->blob.data();
  size_t pos = 4;
  auto fail = [&](const char* why) -> pcc_codec* {
    pcc_set_error("pcc_codec_create: malformed checkpoint (%s)", why);
    pcc_destroy(cd->ctx);
    delete cd;
    return nullptr;
  };
  uint32_t count;
  memcpy(&count, p + pos, 4);
  pos += 4;
  for (uint32_t i = 0; i < count; ++i) {
    if (pos + 2 > n) return fail("truncated");
    uint16_t nl;
    memcpy(&nl, p + pos, 2);
    pos += 2;
    if (pos + nl + 2 > n) return fail("truncated name");
    std::string name((const char*)p + pos, nl);
    pos += nl;
    Tensor t;
    t.dtype = p[pos];
    const int nd = p[pos + 1];
    pos += 2;
    if (pos + 4 * (size_t)nd + 8 > n) return fail("truncated dims");
    for (int d = 0; d < nd; ++d) {
      uint32_t v;
      memcpy(&v, p + pos, 4);
      pos += 4;
      t.dims.push_back(v);
    }
    uint64_t nb;
    memcpy(&nb, p + pos, 8);
    pos += 8;
    pos = (pos + 7) & ~(size_t)7;
    if (pos + nb > n || nb != (uint64_t)t.numel() * 4) return fail("bad tensor size");
    t.data = p + pos;
    t.nbytes = nb;
    pos = (pos + nb + 7) & ~(size_t)7;
    cd->t[name] = t;
  }
  // weights, biases and the small float tables the kernels read go to HBM once
  for (auto& kv : cd->t) {
    const std::string& k = kv.first;
    const bool is_wb = (k.size() > 7 && k.compare(k.size() - 7, 7, ".weight") == 0) ||
                       (k.size() > 5 && k.compare(k.size() - 5, 5, ".bias") == 0);
    const bool is_tab = k == "entropy_bottleneck.medians" || k == "gaussian_conditional.scale_table";
    if ((!is_wb && !is_tab) || k.compare(0, 8, "scale_nn") == 0 || kv.second.dtype != 0) continue;
    float* d = nullptr;
    if (hipMalloc((void**)&d, std::max<size_t>(kv.second.nbytes, 4)) != hipSuccess ||
        hipMemcpy(d, kv.second.data, kv.second.nbytes, hipMemcpyHostToDevice) != hipSuccess) {
      pcc_set_error("pcc_codec_create: upload of '%s' failed", k.c_str());
      pcc_destroy(cd->ctx);
      delete cd;
      return nullptr;
    }
    cd->dev[k] = d;
  }
  // weights of the 32 -> 32 / 32 -> 64 convolutions in matrix-core operand order, once (pcc_conv_prepare); the three
  // generative up stages of g_s additionally with their output columns in the order the following conv reads fastest
  for (auto& kv : cd->t) {
    const std::string& k = kv.first;
    const Tensor& t = kv.second;
    if (k.size() <= 7 || k.compare(k.size() - 7, 7, ".weight") != 0 || t.dtype != 0 || t.dims.size() != 3) continue;
    const std::string layer = k.substr(0, k.size() - 7);
    const bool is_up = layer.find(".up") != std::string::npos;
    // every conv layer whose widths have a matrix-core form (the model default's 32 -> 32 / 64, and any other multiples of
    // 16 a config.yaml may name: pcc_conv_kernel_name)
    if (is_up && t.dims[0] == 8 && t.dims[1] == 32 && t.dims[2] == 32 && cd->dev.count(k))
      (void)pcc_conv_prepare(ctx, cd->dev[k], 8, 32, 32);   // k_convT16 (a failure leaves the per-call copy)
    if (!is_up && (t.dims[0] == 27 || t.dims[0] == 8) && t.dims[1] % 16 == 0 && t.dims[2] % 16 == 0 && t.dims[1] <= 128 &&
        t.dims[2] <= 256 && cd->dev.count(k)) {
      if (pcc_conv_prepare(ctx, cd->dev[k], (int)t.dims[0], (int)t.dims[1], (int)t.dims[2]) != PCC_OK) {
        for (auto& d : cd->dev) (void)hipFree(d.second);
        pcc_destroy(cd->ctx);
        delete cd;
        return nullptr;
      }
    }
    if (is_up && layer.compare(0, 4, "g_s.") == 0 && t.dims[0] == 8 && t.dims[1] == 32 && t.dims[2] == 32 &&
        cd->t.count(layer + ".bias")) {
      std::vector<float> wp((size_t)t.numel()), bp(32);
      const float* bsrc = cd->t[layer + ".bias"].f32();
      for (int64_t r = 0; r < 8 * 32; ++r)
        for (int j = 0; j < 32; ++j) wp[(size_t)r * 32 + j] = t.f32()[r * 32 + pcc_conv16_perm(j)];
      for (int j = 0; j < 32; ++j) bp[j] = bsrc[pcc_conv16_perm(j)];
      float *dw = nullptr, *db = nullptr;
      if (hipMalloc((void**)&dw, wp.size() * 4) != hipSuccess || hipMalloc((void**)&db, 128) != hipSuccess ||
          hipMemcpy(dw, wp.data(), wp.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(db, bp.data(), 128, hipMemcpyHostToDevice) != hipSuccess) {
        pcc_set_error("pcc_codec_create: upload of the permuted '%s' failed", layer.c_str());
        if (dw) (void)hipFree(dw);
        if (db) (void)hipFree(db);
        for (auto& d : cd->dev) (void)hipFree(d.second);
        pcc_destroy(cd->ctx);
        delete cd;
        return nullptr;
      }
      cd->dev[layer + ".weight#perm"] = dw;
      cd->dev[layer + ".bias#perm"] = db;
      (void)pcc_conv_prepare(ctx, dw, 8, 32, 32);   // k_convT16 reads the up stage's weights in operand order too
    }
  }
  static const char* need[] = {"g_a.conv0.weight", "g_a.conv3.weight", "h_a.conv0.weight", "h_s.conv0.weight",
                               "g_s.color.weight", "entropy_bottleneck.quantized_cdf",
                               "gaussian_conditional.quantized_cdf", "gaussian_conditional.scale_table",
                               "entropy_bottleneck.medians", "entropy_model.offsets_ab", "entropy_model.eps"};
  for (const char* k : need)
    if (!cd->t.count(k)) {
      pcc_set_error("pcc_codec_create: checkpoint has no tensor '%s'", k);
      for (auto& kv : cd->dev) (void)hipFree(kv.second);
      pcc_destroy(cd->ctx);
      delete cd;
      return nullptr;
    }
  cd->c_y = (int)cd->t["g_a.conv3.weight"].dims[2];
  cd->c_z = (int)cd->t["entropy_bottleneck.medians"].dims[0];
  {
    const Tensor *gc_cdf = find(cd, "gaussian_conditional.quantized_cdf"), *gc_len = find(cd, "gaussian_conditional.cdf_length"),
                 *gc_off = find(cd, "gaussian_conditional.offset");
    if (gc_cdf && gc_len && gc_off)
      cd->gc_tables = pcc_rans_tables_build(gc_cdf->i32(), (int)gc_cdf->dims[1], gc_len->i32(), gc_off->i32(),
                                            (int)gc_cdf->dims[0]);
    if (!cd->gc_tables) {
      for (auto& kv : cd->dev) (void)hipFree(kv.second);
      pcc_destroy(cd->ctx);
      delete cd;
      return nullptr;
    }
    const Tensor *eb_cdf = find(cd, "entropy_bottleneck.quantized_cdf"), *eb_len = find(cd, "entropy_bottleneck.cdf_length"),
                 *eb_off = find(cd, "entropy_bottleneck.offset");
    cd->gc_dev = pcc_rans_dev_create(gc_cdf->i32(), (int)gc_cdf->dims[1], gc_len->i32(), gc_off->i32(), (int)gc_cdf->dims[0]);
    if (eb_cdf && eb_len && eb_off)
      cd->eb_dev = pcc_rans_dev_create(eb_cdf->i32(), (int)eb_cdf->dims[1], eb_len->i32(), eb_off->i32(), (int)eb_cdf->dims[0]);
    // (a table set too large for the coder's LDS image leaves the pointer null: version 1 is then refused, not faked)
  }
  warm_copy_paths(cd);
  return cd;
}

extern "C" int pcc_codec_set_seek_points(pcc_codec* cd, int pieces) {
  PCC_REQUIRE(cd, PCC_E_ARG, "pcc_codec_set_seek_points: null codec");
  PCC_REQUIRE(pieces == 0 || (pieces >= 2 && pieces <= 64), PCC_E_ARG, "pcc_codec_set_seek_points: %d pieces (0, or 2 .. 64)", pieces);
  cd->seek_points = pieces;
  return PCC_OK;
}

// the seek points of an n-symbol stream cut into `pieces` (oracle/codec_ref.py seek_indexes): multiples of 64 near
// k n / pieces, ascending, inside (0, n); none for streams under 65536 symbols
static void seek_indexes(int64_t n, int pieces, std::vector<int64_t>* out) {
  out->clear();
  if (pieces < 2 || n < 65536) return;
  for (int k = 1; k < pieces; ++k) {
    const int64_t i = (n * k / pieces) & ~(int64_t)63;
    if (i > 0 && i < n && (out->empty() || i > out->back())) out->push_back(i);
  }
}

extern "C" int pcc_codec_set_container_version(pcc_codec* cd, int version) {
  PCC_REQUIRE(cd, PCC_E_ARG, "pcc_codec_set_container_version: null codec");
  PCC_REQUIRE(version == 0 || version == 1, PCC_E_ARG, "pcc_codec_set_container_version: version %d (0 or 1)", version);
  PCC_REQUIRE(version == 0 || (cd->gc_dev && cd->eb_dev), PCC_E_ARG,
              "pcc_codec_set_container_version: the checkpoint's CDF tables have no device form");
  cd->container_version = version;
  return PCC_OK;
}

extern "C" void pcc_codec_destroy(pcc_codec* cd) {
  if (!cd) return;
  if (cd->ctx) (void)pcc_sync(cd->ctx);
  for (auto& kv : cd->dev) (void)hipFree(kv.second);
  for (auto& sl : cd->scale_slot)
    if (sl.dev) (void)hipFree(sl.dev);
  pcc_rans_tables_free(cd->gc_tables);
  pcc_rans_dev_destroy(cd->gc_dev);
  pcc_rans_dev_destroy(cd->eb_dev);
  cd->pool.release();
  if (cd->up_stream) {
    (void)hipStreamSynchronize(cd->up_stream);
    (void)hipStreamDestroy(cd->up_stream);
  }
  if (cd->side_ctx) pcc_destroy(cd->side_ctx);   // synchronises its stream
  if (cd->side_stream) (void)hipStreamDestroy(cd->side_stream);
  if (cd->up_done) (void)hipEventDestroy(cd->up_done);
  for (hipEvent_t e : cd->events) (void)hipEventDestroy(e);
  for (Pinned* p : {&cd->pin_keys, &cd->pin_occ, &cd->pin_zsym, &cd->pin_ysym, &cd->pin_yidx, &cd->pin_flag, &cd->pin_dec,
                    &cd->pin_up})
    p->release();
  pcc_destroy(cd->ctx);
  delete cd;
}

extern "C" pcc_ctx* pcc_codec_ctx(pcc_codec* cd) { return cd ? cd->ctx : nullptr; }

// ---------------------------------------------------------------------------- encode
// The frames of a GOP as the capture stage hands them over (capturer.py:111-126 — per frame points [n_f,3] int16 or
// int32 and colours [n_f,3] float64 or float32, here already in HBM): unpack_batch's concatenation, batch column,
// dtype casts and the (1,r,g,b) feature rows (codec_pipeline.py:243-262, shared/utils.py:10-42) happen inside the
// two kernels that need the values — the Morton keys are formed straight from the frame arrays and the sorted
// feature rows are gathered straight from the colour arrays; no [N,4] staging tensors.
#define PCC_MAX_FRAMES_ARG 32
struct FrameTab {
  const void* pts[PCC_MAX_FRAMES_ARG];
  const void* cols[PCC_MAX_FRAMES_ARG];
  long long off[PCC_MAX_FRAMES_ARG + 1];  // row offsets of the frames in the concatenation
  int nf, pts_i16, cols_f64;
};
// the frames still in host memory (pcc_encode_gop_host_frames): uploaded inside encode_gop_impl, colours behind the sort
struct HostFrames {
  const void* const* pts;
  const void* const* cols;
};

__device__ __forceinline__ int frame_of(const FrameTab& t, long long i) {
  int f = 0;
  while (f + 1 < t.nf && i >= t.off[f + 1]) ++f;
  return f;
}

__global__ __launch_bounds__(256) void k_frames_keys(FrameTab t, int64_t n, uint64_t* __restrict__ keys,
                                                     int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = frame_of(t, i);
  const int64_t j = i - t.off[f];
  int x, y, z;
  if (t.pts_i16) {
    const int16_t* p = (const int16_t*)t.pts[f] + 3 * j;
    x = p[0]; y = p[1]; z = p[2];
  } else {
    const int32_t* p = (const int32_t*)t.pts[f] + 3 * j;
    x = p[0]; y = p[1]; z = p[2];
    const bool bad = (x < -32768) | (x > 32767) | (y < -32768) | (y > 32767) | (z < -32768) | (z > 32767);
    if (bad) atomicOr(flag, 1);
  }
  keys[i] = pcc_morton(f, x, y, z);
}

// first and last sorted key of every frame (frames stay contiguous under the sort: the frame index is the top of the key)
__global__ void k_frame_end_keys(FrameTab t, const uint64_t* __restrict__ keys, uint64_t* __restrict__ out) {
  const int f = threadIdx.x;
  if (f >= t.nf) return;
  const bool any = t.off[f + 1] > t.off[f];
  out[2 * f] = any ? keys[t.off[f]] : 0ull;
  out[2 * f + 1] = any ? keys[t.off[f + 1] - 1] : 0ull;
}

// feature row i of the Morton-sorted tensor = (1, r, g, b) of concatenated row perm[i]
__global__ __launch_bounds__(256) void k_frames_feats(FrameTab t, const uint32_t* __restrict__ perm, int64_t n,
                                                      float4* __restrict__ feats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long src = perm[i];
  const int f = frame_of(t, src);
  const int64_t j = src - t.off[f];
  float r, g, b;
  if (t.cols_f64) {
    const double* c = (const double*)t.cols[f] + 3 * j;
    r = (float)c[0]; g = (float)c[1]; b = (float)c[2];
  } else {
    const float* c = (const float*)t.cols[f] + 3 * j;
    r = c[0]; g = c[1]; b = c[2];
  }
  feats[i] = make_float4(1.0f, r, g, b);
}

// ---- host frames -> HBM (pcc_encode_gop_host_frames).  The frame arrays are pageable numpy memory; handed to
// hipMemcpyAsync as they are, the runtime stages them 1 MB at a time on the calling thread (24 MB of float64 colours:
// 1.0 ms at 24 GB/s, and the GPU waits for them behind the sort).  Here the codec's parked threads write the arrays into
// a pinned staging buffer laid out like the device copy — the colours CONVERTED to float32 on the way (the cast
// unpack_batch does, codec_pipeline.py:243-262; same round-to-nearest conversion as the device's, half the bytes over
// PCIe) — and each piece goes up by DMA on a stream of its own as soon as it is written.
struct UploadPiece {
  size_t off;      // byte offset in the staging region == in the device region
  const void* src;
  size_t count;    // bytes (raw) or elements (f64 -> f32)
  int f64;
};
constexpr int kUploadThreads = 4;
constexpr size_t kUploadPieceBytes = (size_t)1 << 20;  // of staged output

static void upload_pieces(std::vector<UploadPiece>* out, size_t base, const void* src, size_t n_bytes_out, int f64) {
  for (size_t o = 0; o < n_bytes_out; o += kUploadPieceBytes) {
    const size_t len = std::min(kUploadPieceBytes, n_bytes_out - o);
    if (f64)
      out->push_back({base + o, (const double*)src + o / 4, len / 4, 1});
    else
      out->push_back({base + o, (const char*)src + o, len, 0});
  }
}

// runs the pieces on the parked threads: piece i on thread i % kUploadThreads; with `stream` every thread sends its
// piece up as soon as it is staged (dev + off <- staging + off); *status collects the first HIP error
static void upload_run(pcc_codec* cd, const std::vector<UploadPiece>& pieces, char* stage, char* dev, hipStream_t stream,
                       std::atomic<int>* status) {
  cd->workers.ensure(kUploadThreads);
  const int device = cd->device;
  for (int w = 0; w < kUploadThreads; ++w)
    cd->workers.run(w, [&pieces, stage, dev, stream, status, device, w]() {
      if (stream && hipSetDevice(device) != hipSuccess) status->store(1);
      for (size_t i = (size_t)w; i < pieces.size(); i += kUploadThreads) {
        const UploadPiece& p = pieces[i];
        size_t bytes = p.count;
        if (p.f64) {
          float* d = (float*)(stage + p.off);
          const double* sp = (const double*)p.src;
          for (size_t j = 0; j < p.count; ++j) d[j] = (float)sp[j];
          bytes = p.count * 4;
        } else {
          memcpy(stage + p.off, p.src, p.count);
        }
        if (stream && hipMemcpyAsync(dev + p.off, stage + p.off, bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
          status->store(1);
      }
    });
}

static int encode_gop_impl(pcc_codec* cd, const int32_t* d_coords, const float* d_feats, int64_t n, int n_frames,
                           const double* h_q, int n_q, pcc_buf* h_out, int64_t* h_k, double* h_stage_s,
                           const FrameTab* frames_in = nullptr, const HostFrames* host = nullptr) {
  PCC_REQUIRE(cd && cd->ctx, PCC_E_ARG, "pcc_encode_gop: null codec");
  PCC_REQUIRE(n > 0 && (frames_in || (d_coords && d_feats)) && n_frames >= 1 && n_frames <= 65535 && h_q && n_q >= 1 &&
                  n_q <= 64 && h_out,
              PCC_E_ARG, "pcc_encode_gop: bad argument (n=%lld frames=%d q=%d)", (long long)n, n_frames, n_q);
  pcc_ctx* ctx = cd->ctx;
  hipStream_t st = ctx->stream;
  PCC_HIP(hipSetDevice(cd->device));
  PCC_TRY(pcc_sync(ctx));  // the previous call's tensors are dead from here on
  // ... and so are its colour uploads: a call that left early (duplicate coordinates, a coordinate out of range) may have
  // returned with DMAs of cd->up_stream still writing into pool memory
  if (cd->up_stream) PCC_HIP(hipStreamSynchronize(cd->up_stream));
  if (cd->side_stream) PCC_HIP(hipStreamSynchronize(cd->side_stream));   // likewise
  cd->pool.reset();
  cd->events_used = 0;
  cd->sets.clear();
  cd->out.assign((size_t)n_q, {});
  const int cy = cd->c_y, cz = cd->c_z;
  double ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double t0 = now_s();

  // ---- unpack_batch: SparseTensor(coordinates, features) -> Morton-sorted rows
  Feat x;
  std::vector<uint64_t> root_keys;  // first / last input key of every frame, when the frames came as a table
  {
    CODEC_ALLOC(keys, uint64_t, n);
    CODEC_ALLOC(flag, int32_t, 1);
    CODEC_ALLOC(perm, uint32_t, n);
    CODEC_ALLOC(f, float, 4 * n);
    PCC_HIP(hipMemsetAsync(flag, 0, 4, st));
    FrameTab ftab;
    const FrameTab* frames = frames_in;
    std::vector<UploadPiece> col_pieces;
    std::atomic<int> up_status{0};
    struct WaitWorkers {  // the upload jobs read this frame's locals: no way out of the block with one of them running
      PccWorkers* w = nullptr;
      ~WaitWorkers() { if (w) w->wait_all(); }
    } up_guard;
    char* dc = nullptr;
    size_t pts_bytes = 0;
    if (host) {
      // the frame arrays are host memory: points first (6 B per point), keys + sort are queued behind them, and the
      // colours (float32 from here on) cross PCIe on cd->up_stream while the GPU sorts and builds the rule books
      if (!cd->up_stream) {
        PCC_HIP(hipStreamCreateWithFlags(&cd->up_stream, hipStreamNonBlocking));
        PCC_HIP(hipEventCreateWithFlags(&cd->up_done, hipEventDisableTiming));
      }
      PCC_HIP(hipStreamSynchronize(cd->up_stream));  // idle unless the previous call left early
      ftab = *frames_in;
      const size_t pb = ftab.pts_i16 ? 6 : 12, cb = 12;
      char* dp = (char*)cd->pool.alloc((size_t)n * pb + 256 * (size_t)ftab.nf);
      dc = (char*)cd->pool.alloc((size_t)n * cb + 256 * (size_t)ftab.nf);
      if (!dp || !dc) return PCC_E_NOMEM;
      std::vector<UploadPiece> pt_pieces;
      size_t po = 0, co = 0;
      for (int f = 0; f < ftab.nf; ++f) {
        const size_t nf = (size_t)(ftab.off[f + 1] - ftab.off[f]);
        ftab.pts[f] = dp + po;
        upload_pieces(&pt_pieces, po, host->pts[f], nf * pb, 0);
        po += (nf * pb + 255) & ~(size_t)255;
      }
      pts_bytes = po;
      for (int f = 0; f < ftab.nf; ++f) {  // colours staged behind the points
        const size_t nf = (size_t)(ftab.off[f + 1] - ftab.off[f]);
        ftab.cols[f] = dc + co;
        upload_pieces(&col_pieces, co, host->cols[f], nf * cb, ftab.cols_f64);
        co += (nf * cb + 255) & ~(size_t)255;
      }
      ftab.cols_f64 = 0;
      PCC_TRY(cd->pin_up.ensure(pts_bytes + co));
      up_guard.w = &cd->workers;
      upload_run(cd, pt_pieces, (char*)cd->pin_up.p, nullptr, nullptr, &up_status);
      cd->workers.wait_all();
      if (po) PCC_HIP(hipMemcpyAsync(dp, cd->pin_up.p, po, hipMemcpyHostToDevice, st));
      upload_run(cd, col_pieces, (char*)cd->pin_up.p + pts_bytes, dc, cd->up_stream, &up_status);
      frames = &ftab;
    }
    if (frames) {
      hipLaunchKernelGGL(k_frames_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *frames, n, keys, flag);
      PCC_CHECK_LAUNCH();
    } else {
      PCC_TRY(pcc_morton_keys_batch(ctx, d_coords, n, n_frames, keys, flag));   // an index >= n_frames raises the flag
    }
    // Morton keys of int16 coordinates fill bytes 0-5, the frame index the bytes above: the passes are known without a
    // look at the keys (and without the host round trip behind it); a byte that happens to be constant costs one
    // no-op pass
    PCC_TRY(pcc_sort_pairs_bytes(ctx, keys, perm, n, 0x3Fu | (n_frames > 1 ? 0x40u : 0u) | (n_frames > 256 ? 0x80u : 0u)));
    PCC_TRY(cd->pin_flag.ensure(64 + 16 * PCC_MAX_FRAMES_ARG));
    PCC_HIP(hipMemcpyAsync(cd->pin_flag.p, flag, 4, hipMemcpyDeviceToHost, st));
    if (frames) {  // the octree roots of the geometry slots follow from these (octree_root drops the low 9 key bits)
      CODEC_ALLOC(ends, uint64_t, 2 * PCC_MAX_FRAMES_ARG);
      hipLaunchKernelGGL(k_frame_end_keys, dim3(1), dim3(PCC_MAX_FRAMES_ARG), 0, st, *frames, (const uint64_t*)keys, ends);
      PCC_CHECK_LAUNCH();
      PCC_HIP(hipMemcpyAsync(cd->pin_flag.p + 64, ends, (size_t)16 * frames->nf, hipMemcpyDeviceToHost, st));
    }
    // one read-back for the duplicate check and the sizes of the five pyramid levels above the input (g_a: strides
    // 2, 4, 8; h_a: 16, 32), instead of one per level
    int dup = 0;
    std::vector<int64_t> level_n(5, 0);
    PCC_TRY(pcc_level_counts(ctx, keys, n, 0, 5, level_n.data(), &dup));
    PCC_HIP(hipStreamSynchronize(st));
    PCC_REQUIRE(*(int32_t*)cd->pin_flag.p == 0, PCC_E_RANGE,
                "pcc_encode_gop: coordinate outside [-32768,32767] or batch index outside [0, n_frames)");
    PCC_REQUIRE(!dup, PCC_E_DUP, "pcc_encode_gop: duplicate coordinates");
    if (frames) root_keys.assign((const uint64_t*)(cd->pin_flag.p + 64), (const uint64_t*)(cd->pin_flag.p + 64) + 2 * frames->nf);
    x = {new_set(cd, keys, n, 1, n_frames), f, 4};
    x.cs->down_counts = level_n;
    if (host) {
      // the pyramid and the rule book of the input level need coordinates only: queued in front of the feature rows,
      // they run while the last colours are on their way
      int32_t* nbr0;
      PCC_TRY(nbr27_of(cd, x.cs, &nbr0));
      cd->workers.wait_all();
      up_guard.w = nullptr;
      PCC_REQUIRE(up_status.load() == 0, PCC_E_HIP, "pcc_encode_gop_host_frames: upload of the frame arrays failed");
      PCC_HIP(hipEventRecord(cd->up_done, cd->up_stream));
      PCC_HIP(hipStreamWaitEvent(st, cd->up_done, 0));
    }
    if (frames) {
      hipLaunchKernelGGL(k_frames_feats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *frames,
                         (const uint32_t*)perm, n, (float4*)f);
      PCC_CHECK_LAUNCH();
    } else {
      PCC_TRY(pcc_gather_rows(ctx, d_feats, perm, n, 16, f));
    }
  }

  // ---- step 1: analysis g_a + canonical order of y (codec_pipeline.py:270-281)
  std::vector<std::vector<int64_t>> counts(3);
  Feat h = x, y;
  for (int j = 0; j < 3; ++j) {
    const std::vector<int64_t>* offs;
    PCC_TRY(offsets_of(cd, h.cs, &offs));
    for (int f = 0; f < n_frames; ++f) counts[j].push_back((*offs)[f + 1] - (*offs)[f]);
    Feat a, b;
    PCC_TRY(conv3(cd, "g_a.conv" + std::to_string(j), h, 1, &a));
    PCC_TRY(down2(cd, "g_a.down" + std::to_string(j), a, 1, &b));
    h = b;
  }
  PCC_TRY(conv3(cd, "g_a.conv3", h, 0, &y));
  // k[scale][frame]: strides 4, 2, 1 (coarse -> fine)
  const std::vector<int64_t>* kk[3] = {&counts[2], &counts[1], &counts[0]};
  if (h_k)
    for (int s = 0; s < 3; ++s)
      for (int f = 0; f < n_frames; ++f) h_k[s * n_frames + f] = (*kk[s])[f];
  const int64_t ny = y.cs->n;
  View yv;
  PCC_TRY(view_of(cd, y.cs, &yv));
  CODEC_ALLOC(ys_f, float, std::max<int64_t>(ny, 1) * cy);
  if (ny > 0) PCC_TRY(pcc_gather_rows(ctx, y.f, yv.perm, ny, 4 * cy, ys_f));
  const std::vector<int64_t>* yoffs;
  PCC_TRY(offsets_of(cd, y.cs, &yoffs));
  ts[0] = now_s() - t0;

  // ---- step 6: geometry slot of every frame (codec_pipeline.py:441-462).  Nothing on the GPU path needs it, so it is
  // only DEFINED here: it runs on this thread after the y symbols are on their way to the coder threads (below), its
  // kernels queue up behind the quantiser instead of in front of h_a, and its host half (occupancy coding, z string)
  // overlaps the y coders.
  struct FrameGeo {
    int64_t n, occ_off, occ_len;
    int depth;
    int32_t origin[3];
    std::vector<int64_t> level_n;
    std::vector<uint8_t> v2;   // a frame whose blob the device half finished: version 2 (octree2.hip), or version 3 on the synchronous path
    // blob version 3 (PCC_OCTREE_V3_MIN_LEAVES leaves and more): the frame's leaves in K parts under the frame's root
    struct Part {
      int64_t n = 0, occ_off = 0, nodes = 0;
      std::vector<int64_t> level_n;
    };
    int K = 1;
    int counts_at = 0;   // first counts block of the frame (kGeoCounts words each, one per part)
    std::vector<Part> parts;
  };
  std::vector<FrameGeo> geo((size_t)n_frames);
  double geo_dev_s = 0;
  // Asynchronous form — every frame's latent within the single-workgroup octree kernel and the frames' end keys on the
  // host since the sort: the octree kernels are queued and an event recorded, nothing waits.  The kernels write their
  // (small) output — occupancy bytes and level counts — straight into pinned host memory; geometry_finish() waits for
  // the event.
  bool geo_async = false;
  hipEvent_t geo_ev = nullptr;
  std::vector<int64_t> geo_cap((size_t)n_frames, 0);
  constexpr int kGeoCounts = 20;  // uint32 per frame in the counts block (depth + 1 <= 17 used)
  bool geo_small = (int)root_keys.size() == 2 * n_frames;
  for (int f = 0; f < n_frames; ++f) geo_small &= (*yoffs)[f + 1] - (*yoffs)[f] <= pcc_octree_small_max();
  bool geo_queued = false;
  auto geometry_device_half = [&](pcc_ctx* gctx) -> int {   // gctx: where the asynchronous form queues its kernels
    if (geo_queued) return PCC_OK;
    geo_queued = true;
    const double tg = now_s();
    const bool small = geo_small;
    if (small) {
      int64_t cap_total = 0;
      int n_blocks = 0;
      for (int f = 0; f < n_frames; ++f) {
        FrameGeo& g = geo[f];
        g.n = (*yoffs)[f + 1] - (*yoffs)[f];
        g.depth = 0;
        g.occ_off = cap_total;
        g.occ_len = 0;
        g.origin[0] = g.origin[1] = g.origin[2] = 0;
        g.K = g.n >= PCC_OCTREE_V3_MIN_LEAVES ? pcc_octree_parts_for(g.n) : 1;
        g.counts_at = n_blocks;
        n_blocks += g.K;
        if (g.n > 0) {
          octree_root(root_keys[2 * f], root_keys[2 * f + 1], 9, &g.depth, g.origin);
          geo_cap[f] = (g.n * g.depth + 4 * g.K + 4 + 255) & ~(int64_t)255;
          cap_total += geo_cap[f];
        }
      }
      PCC_TRY(cd->pin_occ.ensure((size_t)std::max<int64_t>(cap_total, 1)));
      PCC_TRY(cd->pin_keys.ensure((size_t)kGeoCounts * 4 * n_blocks));
      if (!geo_ev) PCC_TRY(call_event(cd, &geo_ev));
      for (int f = 0; f < n_frames; ++f) {
        FrameGeo& g = geo[f];
        if (g.n == 0) continue;
        // the kernel writes its (small) output straight into the pinned host buffers — they are device-accessible —
        // so the slot needs no transfer of its own
        if (g.K > 1)
          PCC_TRY(pcc_octree_parts_async(gctx, y.cs->keys + (*yoffs)[f], g.n, 9, g.depth, g.K, cd->pin_occ.p + g.occ_off,
                                         geo_cap[f], (uint32_t*)cd->pin_keys.p + kGeoCounts * g.counts_at, kGeoCounts));
        else
          PCC_TRY(pcc_octree_small_async(gctx, y.cs->keys + (*yoffs)[f], g.n, 9, g.depth, cd->pin_occ.p + g.occ_off, geo_cap[f],
                                         (uint32_t*)cd->pin_keys.p + kGeoCounts * g.counts_at));
      }
      PCC_HIP(hipEventRecord(geo_ev, gctx->stream));
      geo_async = true;
      geo_dev_s = now_s() - tg;
      return PCC_OK;
    }
    PCC_TRY(cd->pin_keys.ensure((size_t)std::max<int64_t>(ny, 1) * 8));
    if (ny > 0) PCC_HIP(hipMemcpyAsync(cd->pin_keys.p, y.cs->keys, (size_t)ny * 8, hipMemcpyDeviceToHost, st));
    PCC_HIP(hipStreamSynchronize(st));
    const uint64_t* ykeys_h = (const uint64_t*)cd->pin_keys.p;
    int64_t cap_total = 0;
    for (int f = 0; f < n_frames; ++f) {
      FrameGeo& g = geo[f];
      g.n = (*yoffs)[f + 1] - (*yoffs)[f];
      g.depth = 0;
      g.occ_off = cap_total;
      g.occ_len = 0;
      g.origin[0] = g.origin[1] = g.origin[2] = 0;
      if (g.n > 0) {
        octree_root(ykeys_h[(*yoffs)[f]], ykeys_h[(*yoffs)[f + 1] - 1], 9, &g.depth, g.origin);
        cap_total += g.n * g.depth;
      }
    }
    CODEC_ALLOC(occ, uint8_t, std::max<int64_t>(cap_total, 1));
    PCC_TRY(cd->pin_occ.ensure((size_t)std::max<int64_t>(cap_total, 1)));
    for (int f = 0; f < n_frames; ++f) {
      FrameGeo& g = geo[f];
      if (g.n == 0) continue;
      if (g.n > PCC_OCTREE_V2_MIN_LEAVES) {   // blob version 2: levels AND entropy coder on the GPU
        const int64_t cap2 = 4096 + 17 * g.n;
        int64_t len2 = 0;
        g.v2.resize((size_t)cap2);
        PCC_TRY(pcc_octree2_encode(ctx, y.cs->keys + (*yoffs)[f], g.n, 9, g.depth, g.origin, g.v2.data(), cap2, &len2));
        g.v2.resize((size_t)len2);
        continue;
      }
      if (g.n >= PCC_OCTREE_V3_MIN_LEAVES) {   // blob version 3 on this (synchronous) path: the one-call form
        const int64_t cap3 = 4096 + 3 * g.n * g.depth;
        int64_t len3 = 0;
        g.v2.resize((size_t)cap3);
        PCC_TRY(pcc_octree_encode_version(ctx, y.cs->keys + (*yoffs)[f], g.n, 9, 3, g.v2.data(), cap3, &len3));
        g.v2.resize((size_t)len3);
        continue;
      }
      g.level_n.assign((size_t)g.depth, 0);
      PCC_TRY(pcc_octree_levels(ctx, y.cs->keys + (*yoffs)[f], g.n, 9, g.depth, occ + g.occ_off, g.n * g.depth,
                                g.level_n.data()));
      for (int64_t v : g.level_n) g.occ_len += v;
      PCC_HIP(hipMemcpyAsync(cd->pin_occ.p + g.occ_off, occ + g.occ_off, (size_t)g.occ_len, hipMemcpyDeviceToHost, st));
    }
    PCC_HIP(hipStreamSynchronize(st));  // occupancy bytes and (earlier in the stream) the z symbols are on the host
    geo_dev_s = now_s() - tg;
    return PCC_OK;
  };
  // the device half has arrived (asynchronous form): level counts, and the occupancy bytes the guess did not cover
  auto geometry_arrived = [&]() -> int {
    if (!geo_async) return PCC_OK;
    PCC_HIP(hipEventSynchronize(geo_ev));
    const uint32_t* hc = (const uint32_t*)cd->pin_keys.p;
    for (int f = 0; f < n_frames; ++f) {
      FrameGeo& g = geo[f];
      if (g.n == 0) continue;
      if (g.K > 1) {   // parts: leaf counts first (they place the parts' bytes), then the level counts of each
        g.parts.assign((size_t)g.K, FrameGeo::Part());
        int64_t start = 0;
        for (int k = 0; k < g.K; ++k) {
          const uint32_t* c = hc + kGeoCounts * (g.counts_at + k);
          FrameGeo::Part& pt = g.parts[(size_t)k];
          pt.n = (int64_t)c[g.depth];
          pt.occ_off = g.occ_off + (int64_t)((((uint64_t)start * (uint64_t)g.depth) + 3) & ~(uint64_t)3) + 4 * k;
          pt.level_n.assign((size_t)g.depth, 0);
          for (int L = 0; L < g.depth && pt.n > 0; ++L) {
            pt.level_n[L] = (int64_t)c[L];
            pt.nodes += pt.level_n[L];
          }
          PCC_REQUIRE(start + pt.n <= g.n && pt.nodes <= pt.n * g.depth && (pt.n == 0 || pt.level_n[0] == 1), PCC_E_ARG,
                      "pcc_encode_gop: octree of frame %d, part %d: %lld leaves behind %lld of %lld, %lld nodes", f, k,
                      (long long)pt.n, (long long)start, (long long)g.n, (long long)pt.nodes);
          start += pt.n;
        }
        PCC_REQUIRE(start == g.n, PCC_E_ARG, "pcc_encode_gop: octree of frame %d: its parts hold %lld of %lld leaves", f,
                    (long long)start, (long long)g.n);
        continue;
      }
      g.level_n.assign((size_t)g.depth, 0);
      for (int L = 0; L < g.depth; ++L) {
        g.level_n[L] = (int64_t)hc[kGeoCounts * g.counts_at + L];
        g.occ_len += g.level_n[L];
      }
      PCC_REQUIRE(g.level_n[0] == 1 && g.occ_len <= g.n * g.depth, PCC_E_ARG,
                  "pcc_encode_gop: octree of frame %d: %lld root nodes, %lld nodes for %lld leaves", f,
                  (long long)g.level_n[0], (long long)g.occ_len, (long long)g.n);
    }
    return PCC_OK;
  };

  // Container version 1 codes its strings on the GPU; what does not lie on the way to them runs on the codec's second
  // stream: the octree kernel of the latents here — one workgroup, 58 us — beside h_a, the z string's coder (step 3)
  // beside h_s.  Both had been queued in the compute stream, which they held up for their whole length.
  const bool v1 = cd->container_version == 1;  // y / z strings coded on the GPU (rans_gpu.hip), flagged container
  // (same box, 2 x 30 steps each way: 217.3 / 212.2 frames/s with everything on the compute stream, 222.2 / 224.2 so)
  pcc_ctx* side = nullptr;
  if (v1) PCC_TRY(side_of(cd, &side));
  if (side && geo_small) {
    hipEvent_t ev_y;
    PCC_TRY(call_event(cd, &ev_y));
    PCC_HIP(hipEventRecord(ev_y, st));
    PCC_HIP(hipStreamWaitEvent(side->stream, ev_y, 0));
    PCC_TRY(geometry_device_half(side));
  }

  // ---- step 2: hyper analysis h_a
  t0 = now_s();
  Feat z;
  {
    Feat a, b;
    PCC_TRY(conv3(cd, "h_a.conv0", y, 1, &a));
    PCC_TRY(down2(cd, "h_a.down0", a, 1, &b));
    PCC_TRY(down2(cd, "h_a.down1", b, 0, &z));
  }
  ts[1] = now_s() - t0;

  // ---- step 3: factorized model over the canonically sorted z; z_hat formed on the device
  t0 = now_s();
  const int64_t nz = z.cs->n;
  View zv;
  PCC_TRY(view_of(cd, z.cs, &zv));
  Feat z_hat;
  int32_t* zsym_dev = nullptr;
  {
    CODEC_ALLOC(zs_f, float, std::max<int64_t>(nz, 1) * cz);
    CODEC_ALLOC(zsym, int32_t, std::max<int64_t>(nz, 1) * cz);
    CODEC_ALLOC(zhat_rows, float, std::max<int64_t>(nz, 1) * cz);
    zsym_dev = zsym;
    PCC_TRY(cd->pin_zsym.ensure((size_t)std::max<int64_t>(nz, 1) * cz * 4));
    if (nz > 0) {
      PCC_TRY(pcc_gather_rows(ctx, z.f, zv.perm, nz, 4 * cz, zs_f));
      PCC_TRY(pcc_factorized_quant(ctx, zs_f, nz, cz, cd->dev["entropy_bottleneck.medians"], zsym, zhat_rows));
      if (!v1) PCC_HIP(hipMemcpyAsync(cd->pin_zsym.p, zsym, (size_t)nz * cz * 4, hipMemcpyDeviceToHost, st));
    }
    float* zf;
    PCC_TRY(rows_to_tensor(cd, zv, zhat_rows, cz, &zf));
    z_hat = {z.cs, zf, cz};
  }
  // version 1: room of the coded strings in pinned memory (first attempt: 6 bytes per symbol; the coder's own first
  // attempt holds 1.5 words per symbol), and the z string's coder queued on the second stream
  const int64_t per_y = (int64_t)cy * ny, nzs = (int64_t)cz * nz;
  int64_t cap_y = 0, cap_z = 0;
  long long* d_lens = nullptr;
  hipEvent_t z_done = nullptr;
  if (v1) {
    PCC_REQUIRE(cd->gc_dev && cd->eb_dev, PCC_E_ARG, "pcc_encode_gop: container version 1 without device CDF tables");
    cap_y = std::min<int64_t>(pcc_rans_dev_bound(per_y), (6 * per_y + 65536) / 4 * 4);
    cap_z = std::min<int64_t>(pcc_rans_dev_bound(nzs), (6 * nzs + 65536) / 4 * 4);
    PCC_TRY(cd->pin_ysym.ensure((size_t)cap_y * n_q + (size_t)cap_z + 256));
    d_lens = (long long*)cd->pool.alloc(sizeof(long long) * (size_t)(n_q + 1));
    if (!d_lens) return PCC_E_NOMEM;
    if (side) {
      hipEvent_t ev_z;
      PCC_TRY(call_event(cd, &ev_z));
      PCC_TRY(call_event(cd, &z_done));
      PCC_HIP(hipEventRecord(ev_z, st));
      PCC_HIP(hipStreamWaitEvent(side->stream, ev_z, 0));
      PCC_TRY(pcc_rans_encode_dev_async(side, cd->eb_dev, zsym_dev, nullptr, std::max<int64_t>(nz, 1), nzs, 1,
                                        cd->pin_ysym.p + (size_t)cap_y * n_q, cap_z, d_lens + n_q, 0));
      PCC_HIP(hipEventRecord(z_done, side->stream));
    }
  }
  // host half of the geometry slot + the z string (codec_pipeline.py:294-317): run by this thread next to the y coders
  std::vector<std::vector<uint8_t>> blobs((size_t)n_frames);
  std::vector<uint8_t> z_string;
  double helper_geo_s = 0, helper_z_s = 0;
  const Tensor *eb_cdf = find(cd, "entropy_bottleneck.quantized_cdf"), *eb_len = find(cd, "entropy_bottleneck.cdf_length"),
               *eb_off = find(cd, "entropy_bottleneck.offset");
  PCC_REQUIRE(eb_cdf && eb_len && eb_off, PCC_E_ARG, "pcc_encode_gop: entropy_bottleneck tables missing");
  const bool geo_threads = v1;   // version 0: the codec's threads are coding the y strings while this runs
  auto geometry_finish = [&]() -> int {
    PCC_TRY(geometry_arrived());
    const double t = now_s();
    for (int f = 0; f < n_frames; ++f) {
      FrameGeo& g = geo[f];
      if (!g.v2.empty()) {
        blobs[f].swap(g.v2);
        continue;
      }
      if (g.K > 1 && !g.parts.empty()) {   // blob version 3: the parts' coders side by side where threads are free
        std::vector<std::vector<uint8_t>> pb((size_t)g.K);
        std::vector<int> prc((size_t)g.K, PCC_OK);
        std::vector<std::string> perr((size_t)g.K);
        const int64_t zero = 0;
        const int32_t org0[3] = {0, 0, 0};
        auto pack_part = [&](int k) {
          const FrameGeo::Part& pt = g.parts[(size_t)k];
          pb[(size_t)k].resize((size_t)(64 + 2 * pt.nodes + 16));
          int64_t len = 0;
          prc[(size_t)k] = pcc_octree_pack(pt.n ? cd->pin_occ.p + pt.occ_off : nullptr, pt.n ? pt.level_n.data() : &zero,
                                           pt.n ? g.depth : 0, pt.n, pt.n ? g.origin : org0, pb[(size_t)k].data(),
                                           (int64_t)pb[(size_t)k].size(), &len);
          if (prc[(size_t)k] != PCC_OK) perr[(size_t)k] = pcc_last_error();
          pb[(size_t)k].resize((size_t)(prc[(size_t)k] == PCC_OK ? len : 0));
        };
        if (geo_threads) {   // container version 1: the codec's threads have nothing else to do at this point
          cd->workers.ensure(g.K - 1);
          for (int k = 1; k < g.K; ++k) cd->workers.run(k - 1, [&pack_part, k]() { pack_part(k); });
          pack_part(0);
          cd->workers.wait_all();
        } else {
          for (int k = 0; k < g.K; ++k) pack_part(k);
        }
        for (int k = 0; k < g.K; ++k)
          if (prc[(size_t)k] != PCC_OK) {
            pcc_set_error("pcc_encode_gop (geometry, frame %d part %d): %s", f, k, perr[(size_t)k].c_str());
            return prc[(size_t)k];
          }
        int64_t tot = 64 + 4 * g.K;
        for (const auto& b : pb) tot += (int64_t)b.size();
        blobs[f].resize((size_t)tot);
        int64_t len = 0;
        PCC_TRY(pcc_octree_join_parts(g.depth, g.origin, g.n, pb.data(), g.K, blobs[f].data(), tot, &len));
        blobs[f].resize((size_t)len);
        continue;
      }
      const int64_t cap = 64 + 2 * g.occ_len + 16;
      blobs[f].resize((size_t)cap);
      int64_t len = 0;
      const int64_t zero = 0;
      PCC_TRY(pcc_octree_pack(g.n ? cd->pin_occ.p + g.occ_off : nullptr, g.n ? g.level_n.data() : &zero, g.depth, g.n,
                              g.origin, blobs[f].data(), cap, &len));
      blobs[f].resize((size_t)len);
    }
    helper_geo_s = now_s() - t;
    return PCC_OK;
  };
  auto side_streams = [&]() -> int {
    PCC_TRY(geometry_device_half(ctx));
    PCC_TRY(geometry_finish());
    double t = now_s();
    std::vector<int32_t> idx((size_t)nz * cz);
    for (int c = 0; c < cz; ++c) std::fill(idx.begin() + (size_t)c * nz, idx.begin() + (size_t)(c + 1) * nz, c);
    PCC_TRY(rans_encode_grow((const int32_t*)cd->pin_zsym.p, idx.data(), nz * cz, eb_cdf, eb_len, eb_off, &z_string));
    helper_z_s = now_s() - t;
    return PCC_OK;
  };
  ts[2] = now_s() - t0;

  // ---- step 4: hyper synthesis h_s -> (scales_hat | means_hat) at stride 8
  t0 = now_s();
  Feat gp;
  PCC_TRY(h_s_up(cd, z_hat, &gp));
  ts[3] = now_s() - t0;

  // ---- step 5: all Q quality streams at once (codec_pipeline.py:397-437)
  t0 = now_s();
  std::vector<std::vector<uint8_t>> y_strings((size_t)n_q);  // only the int16-overflow path below fills these
  std::vector<uint8_t> head_done((size_t)n_q, 0);
  std::vector<int64_t> seek_idx;                     // seek points of the y strings (cd->seek_points; version 0 only)
  std::vector<std::vector<uint64_t>> seek_state;     // [quality][point]
  std::vector<std::vector<int64_t>> seek_word;
  if (v1) {
    // Container version 1: symbols and indexes stay in HBM, the Q y streams and the z stream are coded by the GPU's
    // interleaved coder behind the quantiser, and only the finished streams cross PCIe.  The geometry slots (device
    // half + host occupancy coder) follow in the stream / on this thread as in version 0.
    float* params;
    PCC_TRY(h_s_out_at(cd, gp, y.cs, yv, &params));
    float* scale_d;
    PCC_TRY(scale_rows_dev(cd, 0, h_q, n_q, &scale_d));
    const Tensor* tab = find(cd, "gaussian_conditional.scale_table");
    PCC_REQUIRE(tab, PCC_E_ARG, "pcc_encode_gop: gaussian_conditional tables missing");
    const int64_t per = per_y, tot = per * n_q;
    CODEC_ALLOC(sym32, int32_t, std::max<int64_t>(tot, 1));
    CODEC_ALLOC(idx8, uint8_t, std::max<int64_t>(tot, 1));
    if (ny > 0)
      PCC_TRY(pcc_gaussian_quant_dev(ctx, ys_f, params, ny, cy, scale_d, n_q, cd->dev["gaussian_conditional.scale_table"],
                                     (int)tab->dims[0], sym32, idx8));
    PCC_TRY(cd->pin_flag.ensure(8 * 65 + 64));
    long long* h_lens = (long long*)cd->pin_flag.p;
    uint8_t *ystreams = nullptr, *zstream = nullptr;
    PCC_TRY(geometry_device_half(ctx));  // (unless it runs on the second stream already) in front of the coder kernels: its output is on the host while they run
    for (int attempt = 0; attempt < 2; ++attempt) {
      // attempt 0: the room reserved above; attempt 1: the bound
      if (attempt == 1) {
        cap_y = pcc_rans_dev_bound(per);
        cap_z = pcc_rans_dev_bound(nzs);
      }
      // the packing kernels write the finished streams straight into pinned host memory (device-visible): they cross PCIe
      // as they are written, and the one synchronisation for their lengths is also the one for their bytes
      // (same box, three interleaved passes against device buffers + four copies + a second synchronisation: encode
      // 1.98 / 2.10 / 1.72 ms against 2.05 / 2.13 / 1.74)
      PCC_TRY(cd->pin_ysym.ensure((size_t)cap_y * n_q + (size_t)cap_z + 256));
      ystreams = cd->pin_ysym.p;
      zstream = cd->pin_ysym.p + (size_t)cap_y * n_q;
      PCC_TRY(pcc_rans_encode_dev_async(ctx, cd->gc_dev, sym32, idx8, 1, per, n_q, ystreams, cap_y, d_lens, attempt));
      if (attempt == 0 && z_done)   // coded on the second stream since step 3
        PCC_HIP(hipStreamWaitEvent(st, z_done, 0));
      else
        PCC_TRY(pcc_rans_encode_dev_async(ctx, cd->eb_dev, zsym_dev, nullptr, std::max<int64_t>(nz, 1), nzs, 1, zstream, cap_z,
                                          d_lens + n_q, attempt));
      PCC_HIP(hipMemcpyAsync(h_lens, d_lens, (size_t)(n_q + 1) * 8, hipMemcpyDeviceToHost, st));
      if (attempt == 0) PCC_TRY(geometry_finish());   // the occupancy coder of this thread runs while the GPU codes
      PCC_HIP(hipStreamSynchronize(st));
      bool fits = true;
      for (int q = 0; q <= n_q; ++q) fits &= h_lens[q] >= 0;
      if (fits) break;
      PCC_REQUIRE(attempt == 0, PCC_E_NOMEM, "pcc_encode_gop: a coded stream exceeds its bound");
    }
    for (int q = 0; q < n_q; ++q) y_strings[q].assign(ystreams + (size_t)q * cap_y, ystreams + (size_t)q * cap_y + h_lens[q]);
    z_string.assign(zstream, zstream + h_lens[n_q]);
  } else {
    float* params;
    PCC_TRY(h_s_out_at(cd, gp, y.cs, yv, &params));
    float* scale_d;
    PCC_TRY(scale_rows_dev(cd, 0, h_q, n_q, &scale_d));
    const Tensor* tab = find(cd, "gaussian_conditional.scale_table");
    const Tensor *gc_cdf = find(cd, "gaussian_conditional.quantized_cdf"), *gc_len = find(cd, "gaussian_conditional.cdf_length"),
                 *gc_off = find(cd, "gaussian_conditional.offset");
    PCC_REQUIRE(tab && gc_cdf && gc_len && gc_off, PCC_E_ARG, "pcc_encode_gop: gaussian_conditional tables missing");
    const int ntab = (int)tab->dims[0];
    const int64_t per = (int64_t)cy * ny, tot = per * n_q;
    CODEC_ALLOC(sym16, int16_t, std::max<int64_t>(tot, 1));
    CODEC_ALLOC(idx8, uint8_t, std::max<int64_t>(tot, 1));
    CODEC_ALLOC(flag, int32_t, 1);
    PCC_TRY(cd->pin_ysym.ensure((size_t)std::max<int64_t>(tot, 1) * 4));
    PCC_TRY(cd->pin_yidx.ensure((size_t)std::max<int64_t>(tot, 1) * 4));
    PCC_HIP(hipMemsetAsync(flag, 0, 4, st));
    if (ny > 0)
      PCC_TRY(pcc_gaussian_quant16(ctx, ys_f, params, ny, cy, scale_d, n_q, cd->dev["gaussian_conditional.scale_table"],
                                   ntab, sym16, idx8, flag));
    PCC_HIP(hipMemcpyAsync(cd->pin_flag.p, flag, 4, hipMemcpyDeviceToHost, st));
    // The Q streams are coded on Q host threads, and rANS codes from the END of the array: the symbols cross PCIe
    // in kChunks pieces, last piece first (one strided copy moves that piece of every quality), and a coder only
    // waits for the piece it is about to enter — all Q threads start after the first ~2 MB instead of after their
    // whole quality (the last thread used to start 0.2 ms after the quantiser finished).
    const int n_chunks = per >= (1 << 16) ? 4 : 1;
    std::vector<int64_t> bound((size_t)n_chunks);
    for (int c = 0; c < n_chunks; ++c) bound[c] = (per * (n_chunks - 1 - c) / n_chunks) & ~(int64_t)63;
    std::vector<hipEvent_t> evc((size_t)n_chunks, nullptr);
    for (int c = 0; c < n_chunks; ++c) {
      const int64_t lo = bound[c], hi = c == 0 ? per : bound[c - 1];
      if (ny > 0 && hi > lo) {
        PCC_HIP(hipMemcpy2DAsync(cd->pin_ysym.p + (size_t)lo * 2, (size_t)per * 2, sym16 + lo, (size_t)per * 2,
                                 (size_t)(hi - lo) * 2, (size_t)n_q, hipMemcpyDeviceToHost, st));
        PCC_HIP(hipMemcpy2DAsync(cd->pin_yidx.p + (size_t)lo, (size_t)per, idx8 + lo, (size_t)per, (size_t)(hi - lo),
                                 (size_t)n_q, hipMemcpyDeviceToHost, st));
      }
      PCC_TRY(call_event(cd, &evc[c]));
      PCC_HIP(hipEventRecord(evc[c], st));
    }
    struct GateUser {
      hipEvent_t* ev;
      bool failed;
    };
    std::vector<int64_t> lens((size_t)n_q, 0);
    std::vector<uint8_t> stage;
    int rc = PCC_OK;
    int64_t cap = 2 * per + 4096;
    std::vector<int> rcq((size_t)n_q, PCC_OK);
    std::vector<std::string> errq((size_t)n_q);
    seek_indexes(per, v1 ? 0 : cd->seek_points, &seek_idx);
    seek_state.assign((size_t)n_q, std::vector<uint64_t>(seek_idx.size() + 1, 0));
    seek_word.assign((size_t)n_q, std::vector<int64_t>(seek_idx.size() + 1, 0));
    int side_rc = PCC_OK;
    const int64_t nz_for_header = nz;
    auto code_quality = [&](int q) {
      (void)hipSetDevice(cd->device);
      GateUser gu{evc.data(), false};
      PccRansGate gate{n_chunks, bound.data(),
                       [](void* u, int c) {
                         GateUser* g = (GateUser*)u;
                         if (hipEventSynchronize(g->ev[c]) != hipSuccess) g->failed = true;
                       },
                       &gu};
      gate.fn(&gu, 0);  // the overflow flag travels ahead of the first piece
      if (gu.failed) {
        rcq[q] = PCC_E_HIP;
        errq[q] = "hipEventSynchronize failed";
        return;
      }
      if (*(volatile int32_t*)cd->pin_flag.p != 0) return;  // int16 overflow: the generic path below codes everything
      int64_t capq = cap, got = 0;
      std::unique_ptr<uint8_t[]> ybuf;  // not cleared: a std::vector of the worst-case size would memset 1.7 MB first
      for (int attempt = 0; attempt < 2; ++attempt) {
        ybuf.reset(new uint8_t[(size_t)capq]);
        PccRansSeek seek{(int)seek_idx.size(), seek_idx.data(), seek_state[q].data(), seek_word[q].data()};
        rcq[q] = pcc_rans_encode16_seek((const int16_t*)cd->pin_ysym.p + (size_t)q * per,
                                        cd->pin_yidx.p + (size_t)q * per, per, gc_cdf->i32(), (int)gc_cdf->dims[1],
                                        gc_len->i32(), gc_off->i32(), (int)gc_cdf->dims[0], ybuf.get(), capq, &got,
                                        &gate, cd->gc_tables, seek_idx.empty() ? nullptr : &seek);
        if (rcq[q] != PCC_E_NOMEM) break;
        capq = 48 * per + 4096;
      }
      if (rcq[q] == PCC_OK && gu.failed) {
        rcq[q] = PCC_E_HIP;
        errq[q] = "hipEventSynchronize failed";
      } else if (rcq[q] == PCC_OK) {
        // head of the container (codec_pipeline.py:477-499) + y string, written here in parallel with the other
        // qualities; the z string, whose length is patched into the header, and the frame slots follow after the join
        std::vector<uint8_t>& o = cd->out[q];
        o.clear();
        o.reserve((size_t)got + 65536);
        put_be32(o, n_frames);
        put_be_f64(o, h_q[2 * q]);
        put_be_f64(o, h_q[2 * q + 1]);
        put_be32(o, (int32_t)ny);
        put_be32(o, (int32_t)nz_for_header);
        put_be32(o, (int32_t)got);
        put_be32(o, 0);
        o.insert(o.end(), ybuf.get(), ybuf.get() + got);
        head_done[q] = 1;
      } else {
        errq[q] = pcc_last_error();
      }
    };
    {
      struct WaitAll {  // the jobs reference locals of this frame: never leave it while one is running
        PccWorkers& w;
        ~WaitAll() { w.wait_all(); }
      } wait_all{cd->workers};
      cd->workers.ensure(n_q);
      for (int q = 0; q < n_q; ++q) cd->workers.run(q, [&code_quality, q]() { code_quality(q); });
      side_rc = side_streams();  // geometry slots + z string on this thread meanwhile (waits on scope exit)
    }
    PCC_TRY(side_rc);
    for (int q = 0; q < n_q; ++q)
      if (rcq[q] != PCC_OK) {
        pcc_set_error("pcc_encode_gop (y stream %d): %s", q, errq[q].c_str());
        return rcq[q];
      }
    if (*(int32_t*)cd->pin_flag.p == 0) {
      // coded above
    } else {  // a symbol outside int16: the generic int32 form
      std::fill(head_done.begin(), head_done.end(), 0);
      CODEC_ALLOC(sym32, int32_t, std::max<int64_t>(tot, 1));
      CODEC_ALLOC(idx32, int32_t, std::max<int64_t>(tot, 1));
      PCC_TRY(pcc_gaussian_quant(ctx, ys_f, params, ny, cy, scale_d, n_q, cd->dev["gaussian_conditional.scale_table"],
                                 ntab, sym32, idx32));
      PCC_HIP(hipMemcpyAsync(cd->pin_ysym.p, sym32, (size_t)tot * 4, hipMemcpyDeviceToHost, st));
      PCC_HIP(hipMemcpyAsync(cd->pin_yidx.p, idx32, (size_t)tot * 4, hipMemcpyDeviceToHost, st));
      PCC_HIP(hipStreamSynchronize(st));
      for (int attempt = 0; attempt < 2; ++attempt) {
        stage.resize((size_t)cap * n_q);
        rc = pcc_rans_encode_multi((const int32_t*)cd->pin_ysym.p, (const int32_t*)cd->pin_yidx.p, per, n_q,
                                   gc_cdf->i32(), (int)gc_cdf->dims[1], gc_len->i32(), gc_off->i32(),
                                   (int)gc_cdf->dims[0], stage.data(), cap, lens.data());
        if (rc != PCC_E_NOMEM) break;
        cap = 48 * per + 4096;
      }
      PCC_TRY(rc);
      for (int q = 0; q < n_q; ++q)
        y_strings[q].assign(stage.begin() + (size_t)q * cap, stage.begin() + (size_t)q * cap + (size_t)lens[q]);
    }
  }
  ts[5] = now_s() - t0;

  ts[4] = geo_dev_s + helper_geo_s;  // these two ran next to the y coders, inside the gaussian stage's wall time
  ts[2] += helper_z_s;

  // ---- step 7: containers, field for field the reference writer (codec_pipeline.py:464-517)
  t0 = now_s();
  for (int q = 0; q < n_q; ++q) {
    std::vector<uint8_t>& o = cd->out[q];
    if (head_done[q]) {  // header + y string are in place: patch len_z (bytes 32..35, big-endian)
      const uint32_t lz = (uint32_t)z_string.size();
      o[32] = (uint8_t)(lz >> 24); o[33] = (uint8_t)(lz >> 16); o[34] = (uint8_t)(lz >> 8); o[35] = (uint8_t)lz;
    } else {
      o.clear();
      put_be32(o, n_frames | (cd->container_version << 24));
      put_be_f64(o, h_q[2 * q]);
      put_be_f64(o, h_q[2 * q + 1]);
      put_be32(o, (int32_t)ny);
      put_be32(o, (int32_t)nz);
      put_be32(o, (int32_t)y_strings[q].size());
      put_be32(o, (int32_t)z_string.size());
      o.insert(o.end(), y_strings[q].begin(), y_strings[q].end());
    }
    o.insert(o.end(), z_string.begin(), z_string.end());
    for (int f = 0; f < n_frames; ++f) {
      put_be32(o, (int32_t)blobs[f].size());
      for (int s = 0; s < 3; ++s) put_be32(o, (int32_t)(*kk[s])[f]);
      o.insert(o.end(), blobs[f].begin(), blobs[f].end());
    }
    if (head_done[q] && !seek_idx.empty()) {   // "PCSK" | count | count x (index | state | words consumed), big-endian
      const char magic[4] = {'P', 'C', 'S', 'K'};
      o.insert(o.end(), magic, magic + 4);
      put_be32(o, (int32_t)seek_idx.size());
      for (size_t k = 0; k < seek_idx.size(); ++k) {
        put_be32(o, (int32_t)seek_idx[k]);
        put_be32(o, (int32_t)(uint32_t)(seek_state[q][k] >> 32));
        put_be32(o, (int32_t)(uint32_t)seek_state[q][k]);
        put_be32(o, (int32_t)seek_word[q][k]);
      }
    }
    h_out[q].data = o.data();
    h_out[q].len = (int64_t)o.size();
  }
  ts[6] = now_s() - t0;
  if (h_stage_s) memcpy(h_stage_s, ts, 7 * sizeof(double));
  return PCC_OK;
}

// ---------------------------------------------------------------------------- decode
// destinations of pcc_decode_gop_packed: the cloud as pack_batches returns it, queued behind the last kernel
struct PackedDst {
  int32_t* points;
  float* colors;
  int64_t cap;   // rows the two arrays hold
};
__global__ __launch_bounds__(256) void k_pack_cloud(const int4* __restrict__ coords, const float* __restrict__ colors,
                                                    int64_t n, int32_t* __restrict__ xyz, float* __restrict__ rgb);

static int decode_gop_impl(pcc_codec* cd, const uint8_t* h_in, int64_t len, pcc_cloud_info* h_info,
                           double* h_stage_s, const PackedDst* dst = nullptr) {
  PCC_REQUIRE(cd && cd->ctx, PCC_E_ARG, "pcc_decode_gop: null codec");
  PCC_REQUIRE(h_in && len >= 36 && h_info, PCC_E_STREAM, "pcc_decode_gop: container shorter than its header");
  pcc_ctx* ctx = cd->ctx;
  hipStream_t st = ctx->stream;
  PCC_HIP(hipSetDevice(cd->device));
  PCC_TRY(pcc_sync(ctx));
  if (cd->up_stream) PCC_HIP(hipStreamSynchronize(cd->up_stream));   // an encode on this codec that left early (see there)
  if (cd->side_stream) PCC_HIP(hipStreamSynchronize(cd->side_stream));
  cd->pool.reset();
  cd->events_used = 0;
  cd->sets.clear();
  cd->rec_n = 0;
  cd->rec_offsets.clear();
  const int cy = cd->c_y, cz = cd->c_z;
  double ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- step 1: container (codec_parallel.py:173-216)
  double t0 = now_s();
  Reader r{h_in, len};
  // first word: frame count, with the container version in its top byte (0 = the reference's layout; 1 = y / z strings
  // in the GPU coder's interleaved form, rans_gpu.hip)
  const int32_t word0 = r.be32();
  const int version = (int)(((uint32_t)word0 >> 24) & 0xFFu);
  const int32_t n_frames = (int32_t)((uint32_t)word0 & 0x00FFFFFFu);
  PCC_REQUIRE(version == 0 || version == 1, PCC_E_STREAM, "pcc_decode_gop: container version %d", version);
  const bool v1 = version == 1;
  PCC_REQUIRE(!v1 || (cd->gc_dev && cd->eb_dev), PCC_E_ARG, "pcc_decode_gop: version-1 container, but the checkpoint's CDF "
              "tables have no device form");
  const double qg = r.be_f64(), qa = r.be_f64();
  const int32_t ny_hdr = r.be32(), nz_hdr = r.be32(), ylen = r.be32(), zlen = r.be32();
  const uint8_t* ystr = r.bytes(ylen);
  const uint8_t* zstr = r.bytes(zlen);
  PCC_REQUIRE(!r.bad && n_frames >= 0 && n_frames <= 65535 && ny_hdr >= 0 && nz_hdr >= 0, PCC_E_STREAM,
              "pcc_decode_gop: truncated container");
  struct Slot {
    const uint8_t* p;
    int32_t len;
  };
  std::vector<Slot> slots((size_t)n_frames);
  std::vector<int64_t> ks[3];
  for (int f = 0; f < n_frames; ++f) {
    const int32_t pl = r.be32();
    for (int s = 0; s < 3; ++s) ks[s].push_back(r.be32());
    slots[f] = {r.bytes(pl), pl};
    PCC_REQUIRE(!r.bad, PCC_E_STREAM, "pcc_decode_gop: truncated container");
  }
  // Behind the last frame record the reference's reader stops (codec_parallel.py:200-213).  This library's encoder may
  // have left the seek points of the y string there ("PCSK" | count | count x (index | state | words consumed)); they
  // are a hint — every piece decoded from one is checked against the next point, and a trailer that does not parse or
  // does not check out is ignored (the string is then decoded serially, as the reference decodes it)
  std::vector<int64_t> sk_idx, sk_word;
  std::vector<uint64_t> sk_state;
  if (!v1 && len - r.pos >= 8 && memcmp(h_in + r.pos, "PCSK", 4) == 0) {
    Reader t{h_in + r.pos + 4, len - r.pos - 4};
    const int32_t cnt = t.be32();
    bool ok = cnt >= 1 && cnt <= 63 && (int64_t)cnt * 16 <= t.len - t.pos;
    for (int k = 0; ok && k < cnt; ++k) {
      const int64_t i = t.be32();
      const uint64_t hi = (uint32_t)t.be32(), lo = (uint32_t)t.be32();
      const int64_t w = t.be32();
      ok = !t.bad && i > (sk_idx.empty() ? 0 : sk_idx.back()) && i < (int64_t)ny_hdr * cy && w >= 2 && w * 4 <= ylen &&
           ((hi << 32) | lo) >= ((uint64_t)1 << 31);
      sk_idx.push_back(i);
      sk_state.push_back((hi << 32) | lo);
      sk_word.push_back(w);
    }
    if (!ok) sk_idx.clear();
  }
  ts[0] = now_s() - t0;

  // point counts the blobs announce, checked against the header BEFORE anything is sized from them:
  // a rANS symbol costs at least ~2^-16 bit and an octree point at least a fraction of a bit, so a
  // count far beyond what the payload can hold is a corrupt header, not a big frame
  int64_t ny = 0;
  std::vector<int64_t> fn((size_t)n_frames, 0);
  std::vector<int> fdepth((size_t)n_frames, 0);
  for (int f = 0; f < n_frames; ++f) {
    int32_t org[3];
    PCC_TRY(pcc_octree_peek(slots[f].p, slots[f].len, &fn[f], &fdepth[f], org));
    PCC_REQUIRE(fn[f] >= 0 && fn[f] <= ((int64_t)slots[f].len + 64) * 4096, PCC_E_STREAM,
                "pcc_decode_gop: frame %d announces %lld points in a %d-byte blob", f, (long long)fn[f], slots[f].len);
    // The root cube must sit on the 2^depth grid of the biased latent lattice, as octree_root() places it: only then
    // are the octree's upper levels the frame's stride-16 / stride-32 coordinate sets, whose sizes the device path
    // below takes from the geometry decoder without a read-back.  An origin off that grid (a flipped bit in the 12
    // origin bytes passes every other check) would make those counts too small for the kernels that trust them.
    if (fn[f] > 0) {
      const int d = fdepth[f];
      PCC_REQUIRE(d >= 1 && d <= 13, PCC_E_STREAM, "pcc_decode_gop: frame %d: octree depth %d", f, d);
      for (int a = 0; a < 3; ++a) {
        const int64_t biased = (int64_t)org[a] + 4096;
        PCC_REQUIRE(biased >= 0 && biased + ((int64_t)1 << d) <= 8192 && (biased & (((int64_t)1 << d) - 1)) == 0,
                    PCC_E_STREAM, "pcc_decode_gop: frame %d: octree origin %d (axis %d) is not on the 2^%d grid", f,
                    org[a], a, d);
      }
    }
    ny += fn[f];
  }
  PCC_REQUIRE(ny == ny_hdr, PCC_E_STREAM, "pcc_decode_gop: container says N_y=%d, geometry gives %lld", ny_hdr,
              (long long)ny);
  PCC_REQUIRE(nz_hdr <= ny_hdr, PCC_E_STREAM, "pcc_decode_gop: container says N_z=%d > N_y=%d", nz_hdr, ny_hdr);
  PCC_REQUIRE(ny_hdr <= kMaxLatentRows, PCC_E_STREAM, "pcc_decode_gop: container says N_y=%d, this build decodes at most %lld",
              ny_hdr, (long long)kMaxLatentRows);

  // z string: nothing from the GPU is needed, decode it on a helper thread right away
  const Tensor *eb_cdf = find(cd, "entropy_bottleneck.quantized_cdf"), *eb_len = find(cd, "entropy_bottleneck.cdf_length"),
               *eb_off = find(cd, "entropy_bottleneck.offset");
  PCC_REQUIRE(eb_cdf && eb_len && eb_off, PCC_E_ARG, "pcc_decode_gop: entropy_bottleneck tables missing");
  std::vector<int32_t> zsym;
  int z_rc = PCC_OK;
  std::string z_err;
  struct WaitAll {  // the job references locals of this frame: never leave it while the job is running
    PccWorkers& w;
    ~WaitAll() { w.wait_all(); }
  } wait_all{cd->workers};
  cd->workers.ensure(1);
  auto start_z_job = [&]() {
    zsym.resize((size_t)std::max<int64_t>((int64_t)nz_hdr * cz, 1));
    cd->workers.run(0, [&]() {
      if (nz_hdr == 0) return;
      std::vector<int32_t> idx((size_t)nz_hdr * cz);
      for (int c = 0; c < cz; ++c) std::fill(idx.begin() + (size_t)c * nz_hdr, idx.begin() + (size_t)(c + 1) * nz_hdr, c);
      z_rc = pcc_rans_decode(zstr, zlen, idx.data(), (int64_t)nz_hdr * cz, eb_cdf->i32(), (int)eb_cdf->dims[1],
                             eb_len->i32(), eb_off->i32(), (int)eb_cdf->dims[0], zsym.data());
      if (z_rc != PCC_OK) z_err = pcc_last_error();
    });
  };
  // N_z is still only an announced number here.  Up to kEarlyZSymbols symbols (16 MB of buffers) the job starts at once
  // and overlaps the geometry decode; a container announcing more waits until the geometry streams have confirmed N_y.
  const bool z_early = !v1 && (int64_t)nz_hdr * cz <= kEarlyZSymbols;
  if (z_early) start_z_job();
  // version 1: the streams are decoded on the device; their headers are checked here, on the host copy
  int64_t z_steps = 0, z_chunks = 0, y_steps = 0, y_chunks = 0;
  int32_t* d_status = nullptr;
  uint8_t* d_streams_early = nullptr;   // version 1: the strings in HBM and the z symbols, when queued before the geometry
  int32_t* zsym_early = nullptr;
  if (v1) {
    int64_t zn = 0, yn = 0;
    PCC_TRY(pcc_rans_stream_info(zstr, zlen, &zn, &z_steps, &z_chunks));
    PCC_TRY(pcc_rans_stream_info(ystr, ylen, &yn, &y_steps, &y_chunks));
    PCC_REQUIRE(zn == (int64_t)nz_hdr * cz && yn == (int64_t)ny_hdr * cy, PCC_E_STREAM,
                "pcc_decode_gop: streams hold %lld / %lld symbols, header says %lld / %lld", (long long)yn, (long long)zn,
                (long long)ny_hdr * cy, (long long)nz_hdr * cz);
    d_status = (int32_t*)cd->pool.alloc(4);
    if (!d_status) return PCC_E_NOMEM;
    PCC_HIP(hipMemsetAsync(d_status, 0, 4, st));
    // both strings go up now, through pinned memory, behind nothing
    PCC_TRY(cd->pin_dec.ensure((size_t)ylen + (size_t)zlen + 512));
    memcpy(cd->pin_dec.p, zstr, (size_t)zlen);
    memcpy(cd->pin_dec.p + (((size_t)zlen + 255) & ~(size_t)255), ystr, (size_t)ylen);
    // ... and the z stream is decoded at once (its symbol count is what its own header says, checked against the
    // container's above; up to kEarlyZSymbols of them, like the host job of version 0): the kernel runs while this thread
    // decodes the octrees
    if ((int64_t)nz_hdr * cz <= kEarlyZSymbols) {
      const size_t zpad = ((size_t)zlen + 255) & ~(size_t)255;
      CODEC_ALLOC(d_str, uint8_t, zpad + (size_t)ylen + 256);
      CODEC_ALLOC(zs, int32_t, std::max<int64_t>((int64_t)nz_hdr, 1) * cz);
      PCC_HIP(hipMemcpyAsync(d_str, cd->pin_dec.p, zpad + (size_t)ylen, hipMemcpyHostToDevice, st));
      PCC_TRY(pcc_rans_decode_dev(ctx, cd->eb_dev, d_str, zlen, (int64_t)nz_hdr * cz, z_steps, z_chunks, nullptr,
                                  std::max<int64_t>(nz_hdr, 1), zs, d_status));
      d_streams_early = d_str;
      zsym_early = zs;
    }
  }

  // ---- step 2: latent coordinates of every frame (codec_parallel.py:266-289)
  t0 = now_s();
  // Every frame's stream is decoded into a vector that grows with what the stream holds; only after all of them have
  // confirmed their announced counts is anything sized from N_y.
  // The octree's upper levels ARE the stride-16 / stride-32 coordinate sets of the frame (its root cube is aligned to
  // the same power-of-two grid — checked above), so their sizes come out of the geometry decoder and the device never
  // has to report them back; the leaves of an octree are distinct by construction and the range check is done here on
  // the host.
  int n_batch = 0;
  int64_t n16 = 0, n32 = 0;
  std::vector<std::vector<int32_t>> fpts((size_t)n_frames);
  // The rows (frame, 8 x, 8 y, 8 z) of all latents, in pinned memory for the upload.  ny is the sum of the announced
  // counts, bounded above; a frame's rows are written only by what its stream decoded.
  PCC_TRY(cd->pin_keys.ensure((size_t)std::max<int64_t>(ny, 1) * 16));
  int32_t* yc_h = (int32_t*)cd->pin_keys.p;  // [ny,4]
  std::vector<int64_t> frow((size_t)n_frames + 1, 0);
  for (int f = 0; f < n_frames; ++f) frow[(size_t)f + 1] = frow[(size_t)f] + fn[f];
  // blob version 3: the parts of all frames decoded side by side on the codec's threads (worker 0 may hold the z job:
  // workers 1 .. take parts, this thread takes the first of every round).  A job decodes its part and writes its rows
  // itself — the part's leaf count is in its header and the part's decoder holds the stream to it —, so what is left
  // for this thread is the check of the parts against each other.
  struct PartJob {
    int f, k;
    const uint8_t* p;
    int64_t len, n, row0;
    int32_t org[3];
    PccOctPart out;
    bool out_of_range = false;
    int rc = PCC_OK;
    std::string err;
  };
  std::vector<PartJob> pjobs;
  std::vector<int> fpart0((size_t)n_frames, -1), fparts((size_t)n_frames, 0);
  for (int f = 0; f < n_frames; ++f) {
    if (fn[f] == 0 || slots[f].p[1] != 3) continue;
    int K = 0;
    const uint8_t* pp[16];
    int64_t pl[16];
    PCC_TRY(pcc_octree_parts(slots[f].p, slots[f].len, &K, pp, pl));   // the parts' counts add up to fn[f]
    int32_t org[3];
    PCC_TRY(pcc_octree_peek(slots[f].p, slots[f].len, nullptr, nullptr, org));
    fpart0[f] = (int)pjobs.size();
    fparts[f] = K;
    int64_t row = frow[(size_t)f];
    for (int k = 0; k < K; ++k) {
      pjobs.emplace_back();
      PartJob& pj = pjobs.back();
      pj.f = f;
      pj.k = k;
      pj.p = pp[k];
      pj.len = pl[k];
      PCC_TRY(pcc_octree_peek(pp[k], pl[k], &pj.n, nullptr, nullptr));
      pj.row0 = row;
      memcpy(pj.org, org, sizeof(org));
      row += pj.n;
    }
  }
  if (!pjobs.empty()) {
    constexpr int kPartThreads = 8;
    auto run_job = [&pjobs, yc_h](int j) {
      PartJob& pj = pjobs[(size_t)j];
      pj.rc = pcc_octree_unpack_part(pj.p, pj.len, &pj.out);
      if (pj.rc != PCC_OK) {
        pj.err = pcc_last_error();
        return;
      }
      if ((int64_t)pj.out.cells.size() != pj.n) {   // (the part's decoder has checked this against the part's header)
        pj.rc = PCC_E_STREAM;
        pj.err = "part decoded another number of leaves than its header says";
        return;
      }
      int32_t xyz[3 * 64];
      bool bad = false;
      for (int64_t i0 = 0; i0 < pj.n; i0 += 64) {
        const int64_t m = std::min<int64_t>(64, pj.n - i0);
        pcc_octree_cells_to_points(pj.out.cells.data() + i0, m, pj.org, xyz);
        int32_t* dst = yc_h + 4 * (pj.row0 + i0);
        for (int64_t i = 0; i < m; ++i) {
          const int32_t x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
          bad |= (x < -4096) | (x > 4095) | (y < -4096) | (y > 4095) | (z < -4096) | (z > 4095);
          dst[4 * i] = pj.f;
          dst[4 * i + 1] = x * 8;
          dst[4 * i + 2] = y * 8;
          dst[4 * i + 3] = z * 8;
        }
      }
      pj.out_of_range = bad;
    };
    cd->workers.ensure(kPartThreads);
    const int nj = (int)pjobs.size();
    for (int j0 = 0; j0 < nj; j0 += kPartThreads) {
      const int j1 = std::min(nj, j0 + kPartThreads);
      for (int j = j0 + 1; j < j1; ++j) cd->workers.run(j - j0, [&run_job, j]() { run_job(j); });
      run_job(j0);
      cd->workers.wait_range(1, kPartThreads);
    }
    for (const PartJob& pj : pjobs)
      if (pj.rc != PCC_OK) {
        pcc_set_error("pcc_decode_gop (geometry, frame %d part %d): %s", pj.f, pj.k, pj.err.c_str());
        return pj.rc;
      }
  }
  bool out_of_range = false;
  for (int f = 0; f < n_frames; ++f) {
    if (fn[f] == 0) continue;
    int64_t level_n[16];
    if (slots[f].p[1] == 3) {   // the parts against each other: order, counts; the rows are written
      for (int L = 0; L < 16; ++L) level_n[L] = 0;
      bool any = false;
      uint64_t last = 0;
      int64_t total = 0;
      for (int k = 0; k < fparts[f]; ++k) {
        const PartJob& pj = pjobs[(size_t)(fpart0[f] + k)];
        out_of_range |= pj.out_of_range;
        if (pj.out.cells.empty()) continue;
        PCC_REQUIRE(!any || (pj.out.cells.front() >> 6) > (last >> 6), PCC_E_STREAM,
                    "pcc_decode_gop: frame %d: part %d begins inside or in front of the cell the part before it ends in", f, k);
        any = true;
        last = pj.out.cells.back();
        total += (int64_t)pj.out.cells.size();
        for (int L = 0; L < 16; ++L) level_n[L] += pj.out.level_n[L];
      }
      PCC_REQUIRE(total == fn[f], PCC_E_STREAM, "pcc_decode_gop: frame %d decoded %lld points, announced %lld", f, (long long)total,
                  (long long)fn[f]);
    } else {
      if (slots[f].p[1] == 2) {   // blob version 2: decoded by the GPU; fn[f] has passed the plausibility checks above
        int64_t n2 = 0;
        fpts[f].resize((size_t)(3 * fn[f]));
        PCC_TRY(pcc_octree2_decode(ctx, slots[f].p, slots[f].len, nullptr, fpts[f].data(), fn[f], &n2, level_n));
      } else {
        PCC_TRY(pcc_octree_unpack_vec(slots[f].p, slots[f].len, &fpts[f], level_n));
      }
      PCC_REQUIRE((int64_t)fpts[f].size() == 3 * fn[f], PCC_E_STREAM, "pcc_decode_gop: frame %d decoded %zu points, announced %lld",
                  f, fpts[f].size() / 3, (long long)fn[f]);
      const int32_t* pts = fpts[f].data();
      int32_t* dst = yc_h + 4 * frow[(size_t)f];
      for (int64_t i = 0; i < fn[f]; ++i) {
        const int32_t x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        out_of_range |= (x < -4096) | (x > 4095) | (y < -4096) | (y > 4095) | (z < -4096) | (z > 4095);
        dst[4 * i] = f;
        dst[4 * i + 1] = x * 8;
        dst[4 * i + 2] = y * 8;
        dst[4 * i + 3] = z * 8;
      }
    }
    const int depth = fdepth[f];
    n16 += depth >= 1 ? level_n[depth - 1] : 1;
    n32 += depth >= 2 ? level_n[depth - 2] : 1;
    n_batch = f + 1;
  }
  if (!z_early && !v1) start_z_job();
  PCC_REQUIRE(!out_of_range && n_batch <= 65535, PCC_E_RANGE, "pcc_decode_gop: decoded coordinate out of range");
  ts[1] = now_s() - t0;

  // ---- step 3: z coordinates re-derived from the y coordinates, z decoded (codec_parallel.py:291-318)
  t0 = now_s();
  CS* ycs;
  {
    std::vector<int64_t> y_level_n(2, 0);
    CODEC_ALLOC(yc, int32_t, 4 * std::max<int64_t>(ny, 1));
    CODEC_ALLOC(keys, uint64_t, std::max<int64_t>(ny, 1));
    CODEC_ALLOC(flag, int32_t, 1);
    CODEC_ALLOC(perm, uint32_t, std::max<int64_t>(ny, 1));
    if (ny > 0) {
      PCC_HIP(hipMemcpyAsync(yc, yc_h, (size_t)ny * 16, hipMemcpyHostToDevice, st));
      PCC_HIP(hipMemsetAsync(flag, 0, 4, st));
      PCC_TRY(pcc_morton_keys(ctx, yc, ny, keys, flag));
      PCC_TRY(pcc_sort_pairs(ctx, keys, perm, ny, 0));
      y_level_n[0] = n16;  // from the geometry decoder (above): no read-back, the stream keeps running
      y_level_n[1] = n32;
    }
    ycs = new_set(cd, keys, ny, 8, std::max(n_batch, 1));
    if (ny > 0) ycs->down_counts = y_level_n;
  }
  PCC_TRY(down_of(cd, ycs));
  PCC_TRY(down_of(cd, ycs->down));
  CS* zcs = ycs->down->down;
  View zv;
  PCC_TRY(view_of(cd, zcs, &zv));
  PCC_REQUIRE(zcs->n == nz_hdr, PCC_E_STREAM, "pcc_decode_gop: container says N_z=%d, coordinates give %lld", nz_hdr,
              (long long)zcs->n);
  cd->workers.wait_all();
  if (z_rc != PCC_OK) {
    pcc_set_error("pcc_decode_gop (z stream): %s", z_err.c_str());
    return z_rc;
  }
  Feat z_hat;
  uint8_t* d_ystr = nullptr;
  {
    const int64_t nz = zcs->n;
    CODEC_ALLOC(zsym_d_own, int32_t, std::max<int64_t>(nz, 1) * cz);
    int32_t* zsym_d = zsym_d_own;
    CODEC_ALLOC(rows, float, std::max<int64_t>(nz, 1) * cz);
    if (v1) {
      const size_t zpad = ((size_t)zlen + 255) & ~(size_t)255;
      if (zsym_early) {  // (nz == nz_hdr was checked above)
        d_ystr = d_streams_early + zpad;
        zsym_d = zsym_early;
      } else {
        CODEC_ALLOC(d_str, uint8_t, zpad + (size_t)ylen + 256);
        PCC_HIP(hipMemcpyAsync(d_str, cd->pin_dec.p, zpad + (size_t)ylen, hipMemcpyHostToDevice, st));
        d_ystr = d_str + zpad;
        PCC_TRY(pcc_rans_decode_dev(ctx, cd->eb_dev, d_str, zlen, nz * cz, z_steps, z_chunks, nullptr, std::max<int64_t>(nz, 1),
                                    zsym_d, d_status));
      }
      if (nz > 0) PCC_TRY(pcc_factorized_dequant(ctx, zsym_d, nz, cz, cd->dev["entropy_bottleneck.medians"], rows));
    } else if (nz > 0) {
      PCC_TRY(cd->pin_zsym.ensure((size_t)nz * cz * 4));
      memcpy(cd->pin_zsym.p, zsym.data(), (size_t)nz * cz * 4);
      PCC_HIP(hipMemcpyAsync(zsym_d, cd->pin_zsym.p, (size_t)nz * cz * 4, hipMemcpyHostToDevice, st));
      PCC_TRY(pcc_factorized_dequant(ctx, zsym_d, nz, cz, cd->dev["entropy_bottleneck.medians"], rows));
    }
    float* zf;
    PCC_TRY(rows_to_tensor(cd, zv, rows, cz, &zf));
    z_hat = {zcs, zf, cz};
  }
  ts[2] = now_s() - t0;

  // ---- step 4: hyper synthesis
  t0 = now_s();
  Feat gp;
  PCC_TRY(h_s_up(cd, z_hat, &gp));
  ts[3] = now_s() - t0;

  // ---- step 5: decode y, de-quantise with offsets (codec_parallel.py:382-419)
  t0 = now_s();
  Feat y_hat;
  {
    View yv;
    PCC_TRY(view_of(cd, ycs, &yv));
    float* params;
    PCC_TRY(h_s_out_at(cd, gp, ycs, yv, &params));
    float* scale_d;
    const double q_dec[2] = {qg, qa};
    PCC_TRY(scale_rows_dev(cd, 1, q_dec, 1, &scale_d));
    const Tensor* tab = find(cd, "gaussian_conditional.scale_table");
    const Tensor *gc_cdf = find(cd, "gaussian_conditional.quantized_cdf"), *gc_len = find(cd, "gaussian_conditional.cdf_length"),
                 *gc_off = find(cd, "gaussian_conditional.offset");
    const Tensor* ab = find(cd, "entropy_model.offsets_ab");
    PCC_REQUIRE(tab && gc_cdf && gc_len && gc_off && ab, PCC_E_ARG, "pcc_decode_gop: gaussian_conditional tables missing");
    const int64_t tot = (int64_t)cy * ny;
    CODEC_ALLOC(idx8, uint8_t, std::max<int64_t>(tot, 1));
    CODEC_ALLOC(sym_d, int32_t, std::max<int64_t>(tot, 1));
    CODEC_ALLOC(rows, float, std::max<int64_t>(tot, 1));
    if (ny > 0 && v1) {
      // version 1: indexes, stream and symbols never leave the device, and the host does not wait for anything here
      PCC_TRY(pcc_gaussian_indexes8(ctx, params, ny, cy, scale_d, cd->dev["gaussian_conditional.scale_table"],
                                    (int)tab->dims[0], idx8));
      PCC_TRY(pcc_rans_decode_dev(ctx, cd->gc_dev, d_ystr, ylen, tot, y_steps, y_chunks, idx8, 1, sym_d, d_status));
      PCC_TRY(pcc_gaussian_dequant(ctx, sym_d, params, ny, cy, scale_d, tab->f32()[0], ab->f32()[0], ab->f32()[1], rows));
    } else if (ny > 0) {
      PCC_TRY(pcc_gaussian_indexes8(ctx, params, ny, cy, scale_d, cd->dev["gaussian_conditional.scale_table"],
                                    (int)tab->dims[0], idx8));
      PCC_TRY(cd->pin_yidx.ensure((size_t)tot));
      PCC_TRY(cd->pin_dec.ensure((size_t)tot * 4));
      PCC_HIP(hipMemcpyAsync(cd->pin_yidx.p, idx8, (size_t)tot, hipMemcpyDeviceToHost, st));
      PCC_HIP(hipStreamSynchronize(st));
      {  // coordinates and rule book of the first synthesis stage do not depend on the y values:
         // the GPU builds them while the host decodes the y stream
        CS* c0;
        int32_t* nbr0;
        PCC_TRY(up_of(cd, ycs, &c0));
        PCC_TRY(nbr27_of(cd, pcc_conv_up_fused() ? ycs : c0, &nbr0));
      }
      // seek points: the pieces between them on host threads, each checked against the point behind it
      bool decoded = false;
      if (!sk_idx.empty() && cd->gc_tables) {
        const int pieces = (int)sk_idx.size() + 1;
        std::vector<int> prc((size_t)pieces, PCC_OK);
        std::vector<uint64_t> end_x((size_t)pieces, 0);
        std::vector<int64_t> end_w((size_t)pieces, 0);
        auto piece = [&](int k) {
          const int64_t lo = k == 0 ? 0 : sk_idx[k - 1], hi = k == pieces - 1 ? tot : sk_idx[k];
          prc[k] = pcc_rans_decode8_range(ystr, ylen, cd->pin_yidx.p, tot, gc_cdf->i32(), (int)gc_cdf->dims[1], gc_len->i32(),
                                          gc_off->i32(), (int)gc_cdf->dims[0], (int32_t*)cd->pin_dec.p, cd->gc_tables, lo, hi,
                                          k == 0 ? 0 : sk_state[k - 1], k == 0 ? 0 : sk_word[k - 1], &end_x[k], &end_w[k]);
        };
        {
          struct WaitPieces {
            PccWorkers& w;
            ~WaitPieces() { w.wait_all(); }
          } wait_pieces{cd->workers};
          cd->workers.ensure(pieces - 1);
          for (int k = 1; k < pieces; ++k) cd->workers.run(k - 1, [&piece, k]() { piece(k); });
          piece(0);
        }
        decoded = true;
        for (int k = 0; k < pieces; ++k) {
          decoded &= prc[k] == PCC_OK;
          if (k + 1 < pieces) decoded &= end_x[k] == sk_state[k] && end_w[k] == sk_word[k];
        }
        if (decoded)
          PCC_HIP(hipMemcpyAsync(sym_d, cd->pin_dec.p, (size_t)tot * 4, hipMemcpyHostToDevice, st));
      }
      // the decoded symbols go back to the device in pieces while the host is still decoding the next piece
      const int n_chunks = tot >= (1 << 16) ? 4 : 1;
      std::vector<int64_t> bound((size_t)n_chunks);
      for (int c = 0; c < n_chunks; ++c) bound[c] = c == n_chunks - 1 ? tot : (tot * (c + 1) / n_chunks) & ~(int64_t)63;
      struct UpUser {
        const int64_t* bound;
        const uint8_t* host;
        int32_t* dev;
        hipStream_t st;
        bool failed;
      } up{bound.data(), cd->pin_dec.p, sym_d, st, false};
      PccRansGate gate{n_chunks, bound.data(),
                       [](void* u, int c) {
                         UpUser* g = (UpUser*)u;
                         const int64_t lo = c == 0 ? 0 : g->bound[c - 1], hi = g->bound[c];
                         if (hi > lo && hipMemcpyAsync(g->dev + lo, g->host + (size_t)lo * 4, (size_t)(hi - lo) * 4,
                                                       hipMemcpyHostToDevice, g->st) != hipSuccess)
                           g->failed = true;
                       },
                       &up};
      if (!decoded)
        PCC_TRY(pcc_rans_decode8_gated(ystr, ylen, cd->pin_yidx.p, tot, gc_cdf->i32(), (int)gc_cdf->dims[1],
                                       gc_len->i32(), gc_off->i32(), (int)gc_cdf->dims[0], (int32_t*)cd->pin_dec.p,
                                       &gate, cd->gc_tables));
      PCC_REQUIRE(!up.failed, PCC_E_HIP, "pcc_decode_gop: hipMemcpyAsync of decoded symbols failed");
      PCC_TRY(pcc_gaussian_dequant(ctx, sym_d, params, ny, cy, scale_d, tab->f32()[0], ab->f32()[0], ab->f32()[1], rows));
    }
    float* yf;
    PCC_TRY(rows_to_tensor(cd, yv, rows, cy, &yf));
    y_hat = {ycs, yf, cy};
  }
  ts[4] = now_s() - t0;

  // ---- step 6: synthesis g_s with per-frame top-k pruning (codec_parallel.py:465-472)
  t0 = now_s();
  Feat h = y_hat;
  const int nb = h.cs->n_batch;
  {
    const std::vector<int64_t>* o0;
    PCC_TRY(offsets_of(cd, h.cs, &o0));
  }
  const float* rgb_cand = nullptr;  // colours of the last stage's candidate rows, when its conv evaluated the colour head
  for (int j = 0; j < 3; ++j) {
    const std::string uname = "g_s.up" + std::to_string(j);
    const std::string cname = "g_s.conv" + std::to_string(j), oname = "g_s.occ" + std::to_string(j);
    const float *w, *b, *hw, *hb;
    const Tensor *tw, *thw;
    PCC_TRY(wb(cd, cname, &w, &b, &tw));
    PCC_TRY(wb(cd, oname, &hw, &hb, &thw));
    const int cin = (int)tw->dims[1], cout = (int)tw->dims[2];
    // the conv that forms its rule book in-kernel reads candidate rows stored channel-permuted: the up stage writes them so
    const bool fused = pcc_conv_up_fused() && cin == 32 && cout == 32 && cd->dev.count(uname + ".weight#perm");
    Feat u;
    PCC_TRY(up2(cd, uname, h, 1, &u, fused));
    const int64_t nu = u.cs->n;
    // last stage: only the occupancy logit and the colour of a candidate row are ever read again, so the conv evaluates
    // the colour head on every candidate row too and does not store the rows (128 B each) at all
    const Tensor* tcol = find(cd, "g_s.color.weight");
    const bool with_rgb = fused && j == 2 && nu > 0 && tcol && tcol->dims.size() == 2 && tcol->dims[0] == 32 &&
                          tcol->dims[1] == 3;
    float* feats = nullptr;
    if (!with_rgb) {
      feats = (float*)cd->pool.alloc(sizeof(float) * (size_t)(std::max<int64_t>(nu, 1) * cout));
      if (!feats) return PCC_E_NOMEM;
    }
    CODEC_ALLOC(logits, float, std::max<int64_t>(nu, 1));
    if (with_rgb) {
      int32_t* pn;
      PCC_TRY(nbr27_of(cd, h.cs, &pn));
      CODEC_ALLOC(rgb_all, float, nu * 3);
      rgb_cand = rgb_all;
      PCC_TRY(pcc_sparse_conv_head_up_perm_rgb(ctx, u.f, h.cs->n, pn, h.cs->n, w, b, 1, hw, hb, logits,
                                               cd->dev["g_s.color.weight"], cd->dev["g_s.color.bias"], rgb_all));
    } else if (fused && nu > 0) {
      // rule book of the 8N candidates formed inside the conv from the book of the N rows below
      int32_t* pn;
      PCC_TRY(nbr27_of(cd, h.cs, &pn));
      PCC_TRY(pcc_sparse_conv_head_up_perm(ctx, u.f, h.cs->n, pn, h.cs->n, w, b, 1, feats, hw, hb, logits));
    } else if (nu == 0) {
      // nothing to convolve
    } else {
      int32_t* nbr;
      PCC_TRY(nbr27_of(cd, u.cs, &nbr));
      PCC_TRY(pcc_sparse_conv_head(ctx, u.f, nu, nbr, 27, nu, nu, w, b, cin, cout, 1, feats, hw, hb, logits));
    }
    const std::vector<int64_t>* offs;
    PCC_TRY(offsets_of(cd, u.cs, &offs));
    std::vector<int64_t> kj((size_t)nb), new_offs(1, 0);
    for (int f = 0; f < nb; ++f) {
      const int64_t want = f < (int)ks[j].size() ? ks[j][f] : 0;
      kj[f] = std::max<int64_t>(0, std::min<int64_t>(want, (*offs)[f + 1] - (*offs)[f]));
      new_offs.push_back(new_offs.back() + kj[f]);
    }
    CODEC_ALLOC(keep, uint32_t, std::max<int64_t>(nu, 1));
    // exact top-k keeps sum_f kj[f] rows: no count to read back, the stage stays asynchronous
    const int64_t n_keep = new_offs.back();
    // the stages that are followed by another want the rule book of what they keep (nbr27_of): the placement writes the
    // candidates' positions among the kept rows as it goes
    int32_t* keep_remap = nullptr;
    if (nu > 0 && j + 1 < 3) {
      keep_remap = (int32_t*)cd->pool.alloc(sizeof(int32_t) * (size_t)nu);
      if (!keep_remap) return PCC_E_NOMEM;
    }
    if (nu > 0) PCC_TRY(pcc_topk_prune_map(ctx, logits, nu, nb, offs->data(), kj.data(), keep, nullptr, keep_remap));
    CODEC_ALLOC(pkeys, uint64_t, std::max<int64_t>(n_keep, 1));
    CODEC_ALLOC(pf, float, std::max<int64_t>(n_keep, 1) * cout);
    // the kept rows stay where they are: the next up stage / the colour head read them through `keep`
    const bool in_place = cout == 32;
    if (n_keep > 0) {
      if (!u.cs->keys && u.cs->gen_parent) {   // the kept candidates' keys from their parents': the 8N are never written
        uint64_t* pk;
        PCC_TRY(keys_of(cd, u.cs->gen_parent, &pk));
        PCC_TRY(pcc_up_coords_rows(ctx, pk, u.cs->gen_parent->n, 3 * log2i(u.cs->stride), keep, n_keep, pkeys));
      } else {
        PCC_TRY(pcc_gather_rows(ctx, u.cs->keys, keep, n_keep, 8, pkeys));
      }
      if (!in_place) PCC_TRY(pcc_gather_rows(ctx, feats, keep, n_keep, 4 * cout, pf));
    }
    CS* ps = new_set(cd, pkeys, n_keep, u.cs->stride, nb);
    ps->offsets = new_offs;
    ps->subset_of = u.cs;
    ps->keep = keep;
    ps->keep_remap = keep_remap;
    h = {ps, in_place ? feats : pf, cout, in_place ? keep : nullptr};
  }
  {
    const float *w, *b;
    const Tensor* tw;
    PCC_TRY(wb(cd, "g_s.color", &w, &b, &tw));
    const int64_t nr = h.cs->n;
    CODEC_ALLOC(rgb, float, std::max<int64_t>(nr, 1) * 3);
    CODEC_ALLOC(coords, int32_t, std::max<int64_t>(nr, 1) * 4);
    if (nr > 0) {
      if (rgb_cand && h.rows)
        PCC_TRY(pcc_gather_rows(ctx, rgb_cand, h.rows, nr, 12, rgb));
      else if (h.rows && (int)tw->dims[0] == 32 && (int)tw->dims[1] <= 8)
        PCC_TRY(pcc_linear_gather(ctx, h.f, h.rows, nr, w, b, (int)tw->dims[1], 0, rgb));
      else if (h.rows) {  // a head the fused form does not cover: gather first
        CODEC_ALLOC(pf, float, nr * h.c);
        PCC_TRY(pcc_gather_rows(ctx, h.f, h.rows, nr, 4 * h.c, pf));
        PCC_TRY(pcc_linear(ctx, pf, nr, w, b, (int)tw->dims[0], (int)tw->dims[1], 0, rgb));
      } else
        PCC_TRY(pcc_linear(ctx, h.f, nr, w, b, (int)tw->dims[0], (int)tw->dims[1], 0, rgb));
      uint64_t* hkeys;
      PCC_TRY(keys_of(cd, h.cs, &hkeys));
      PCC_TRY(pcc_keys_to_coords(ctx, hkeys, nr, coords));
    }
    cd->rec_coords = coords;
    cd->rec_colors = rgb;
    cd->rec_n = nr;
    cd->rec_offsets = h.cs->offsets;
  }
  if (v1) {
    PCC_TRY(cd->pin_flag.ensure(64));
    PCC_HIP(hipMemcpyAsync(cd->pin_flag.p, d_status, 4, hipMemcpyDeviceToHost, st));
  }
  if (dst && cd->rec_n > 0) {
    // pcc_decode_fetch_packed's work, queued here: no synchronisation and no trip through the caller in between
    const int64_t n = cd->rec_n;
    PCC_REQUIRE(n <= dst->cap && dst->points && dst->colors, PCC_E_ARG,
                "pcc_decode_gop_packed: %lld points decoded, destination holds %lld", (long long)n, (long long)dst->cap);
    CODEC_ALLOC(xyz, int32_t, 3 * n);
    CODEC_ALLOC(rgb3, float, 3 * n);
    hipLaunchKernelGGL(k_pack_cloud, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int4*)cd->rec_coords,
                       (const float*)cd->rec_colors, n, xyz, rgb3);
    PCC_CHECK_LAUNCH();
    PCC_HIP(hipMemcpyAsync(dst->points, xyz, (size_t)n * 12, hipMemcpyDefault, st));
    PCC_HIP(hipMemcpyAsync(dst->colors, rgb3, (size_t)n * 12, hipMemcpyDefault, st));
  }
  PCC_TRY(pcc_sync(ctx));
  if (v1 && *(volatile int32_t*)cd->pin_flag.p != 0) {
    cd->rec_n = 0;
    cd->rec_offsets.clear();
    pcc_set_error("pcc_decode_gop: malformed interleaved rANS stream (status %d)", *(int32_t*)cd->pin_flag.p);
    return PCC_E_STREAM;
  }
  ts[5] = now_s() - t0;

  h_info->n_points = cd->rec_n;
  h_info->n_frames = n_frames;
  h_info->n_offsets = (int32_t)cd->rec_offsets.size();
  h_info->h_offsets = cd->rec_offsets.data();
  h_info->d_coords = cd->rec_coords;
  h_info->d_colors = cd->rec_colors;
  h_info->q_g = qg;
  h_info->q_a = qa;
  if (h_stage_s) memcpy(h_stage_s, ts, 6 * sizeof(double));
  return PCC_OK;
}

// the library never throws across the ABI: allocation failures of the host containers become codes
extern "C" int pcc_encode_gop(pcc_codec* cd, const int32_t* d_coords, const float* d_feats, int64_t n,
                              int n_frames, const double* h_q, int n_q, pcc_buf* h_out, int64_t* h_k,
                              double* h_stage_s) {
  try {
    return encode_gop_impl(cd, d_coords, d_feats, n, n_frames, h_q, n_q, h_out, h_k, h_stage_s);
  } catch (const std::bad_alloc&) {
    pcc_set_error("pcc_encode_gop: out of host memory");
    return PCC_E_NOMEM;
  } catch (const std::exception& e) {
    pcc_set_error("pcc_encode_gop: %s", e.what());
    return PCC_E_ARG;
  }
}

extern "C" int pcc_encode_gop_frames(pcc_codec* cd, const void* const* h_d_points, int points_i16,
                                     const void* const* h_d_colors, int colors_f64, const int64_t* h_n, int n_frames,
                                     const double* h_q, int n_q, pcc_buf* h_out, int64_t* h_k, double* h_stage_s) {
  PCC_REQUIRE(h_d_points && h_d_colors && h_n && n_frames >= 1, PCC_E_ARG, "pcc_encode_gop_frames: null frame table");
  PCC_REQUIRE(n_frames <= PCC_MAX_FRAMES_ARG, PCC_E_ARG,
              "pcc_encode_gop_frames: %d frames, at most %d per call (concatenate and use pcc_encode_gop)", n_frames,
              PCC_MAX_FRAMES_ARG);
  FrameTab t;
  memset(&t, 0, sizeof(t));
  t.nf = n_frames;
  t.pts_i16 = points_i16 ? 1 : 0;
  t.cols_f64 = colors_f64 ? 1 : 0;
  for (int f = 0; f < n_frames; ++f) {
    PCC_REQUIRE(h_n[f] >= 0 && (h_n[f] == 0 || (h_d_points[f] && h_d_colors[f])), PCC_E_ARG,
                "pcc_encode_gop_frames: frame %d: n=%lld or null arrays", f, (long long)h_n[f]);
    t.pts[f] = h_d_points[f];
    t.cols[f] = h_d_colors[f];
    t.off[f + 1] = t.off[f] + h_n[f];
  }
  const int64_t n = t.off[n_frames];
  PCC_REQUIRE(n < ((int64_t)1 << 32), PCC_E_ARG, "pcc_encode_gop_frames: %lld points", (long long)n);
  try {
    return encode_gop_impl(cd, nullptr, nullptr, n, n_frames, h_q, n_q, h_out, h_k, h_stage_s, &t);
  } catch (const std::bad_alloc&) {
    pcc_set_error("pcc_encode_gop_frames: out of host memory");
    return PCC_E_NOMEM;
  } catch (const std::exception& e) {
    pcc_set_error("pcc_encode_gop_frames: %s", e.what());
    return PCC_E_ARG;
  }
}

extern "C" int pcc_encode_gop_host_frames(pcc_codec* cd, const void* const* h_points, int points_i16,
                                          const void* const* h_colors, int colors_f64, const int64_t* h_n, int n_frames,
                                          const double* h_q, int n_q, pcc_buf* h_out, int64_t* h_k, double* h_stage_s) {
  PCC_REQUIRE(h_points && h_colors && h_n && n_frames >= 1, PCC_E_ARG, "pcc_encode_gop_host_frames: null frame table");
  PCC_REQUIRE(n_frames <= PCC_MAX_FRAMES_ARG, PCC_E_ARG, "pcc_encode_gop_host_frames: %d frames, at most %d per call",
              n_frames, PCC_MAX_FRAMES_ARG);
  FrameTab t;
  memset(&t, 0, sizeof(t));
  t.nf = n_frames;
  t.pts_i16 = points_i16 ? 1 : 0;
  t.cols_f64 = colors_f64 ? 1 : 0;
  for (int f = 0; f < n_frames; ++f) {
    PCC_REQUIRE(h_n[f] >= 0 && (h_n[f] == 0 || (h_points[f] && h_colors[f])), PCC_E_ARG,
                "pcc_encode_gop_host_frames: frame %d: n=%lld or null arrays", f, (long long)h_n[f]);
    t.off[f + 1] = t.off[f] + h_n[f];
  }
  const int64_t n = t.off[n_frames];
  PCC_REQUIRE(n < ((int64_t)1 << 32), PCC_E_ARG, "pcc_encode_gop_host_frames: %lld points", (long long)n);
  const HostFrames host{h_points, h_colors};
  try {
    return encode_gop_impl(cd, nullptr, nullptr, n, n_frames, h_q, n_q, h_out, h_k, h_stage_s, &t, &host);
  } catch (const std::bad_alloc&) {
    pcc_set_error("pcc_encode_gop_host_frames: out of host memory");
    return PCC_E_NOMEM;
  } catch (const std::exception& e) {
    pcc_set_error("pcc_encode_gop_host_frames: %s", e.what());
    return PCC_E_ARG;
  }
}

extern "C" int pcc_decode_gop(pcc_codec* cd, const uint8_t* h_in, int64_t len, pcc_cloud_info* h_info,
                              double* h_stage_s) {
  try {
    return decode_gop_impl(cd, h_in, len, h_info, h_stage_s);
  } catch (const std::bad_alloc&) {
    pcc_set_error("pcc_decode_gop: out of host memory");
    return PCC_E_NOMEM;
  } catch (const std::exception& e) {
    pcc_set_error("pcc_decode_gop: %s", e.what());
    return PCC_E_STREAM;
  }
}

extern "C" int pcc_decode_gop_packed(pcc_codec* cd, const uint8_t* h_in, int64_t len, int32_t* points, float* colors,
                                     int64_t cap_points, pcc_cloud_info* h_info, double* h_stage_s) {
  PCC_REQUIRE(points && colors && cap_points >= 0, PCC_E_ARG, "pcc_decode_gop_packed: null destination");
  const PackedDst dst{points, colors, cap_points};
  try {
    return decode_gop_impl(cd, h_in, len, h_info, h_stage_s, &dst);
  } catch (const std::bad_alloc&) {
    pcc_set_error("pcc_decode_gop_packed: out of host memory");
    return PCC_E_NOMEM;
  } catch (const std::exception& e) {
    pcc_set_error("pcc_decode_gop_packed: %s", e.what());
    return PCC_E_STREAM;
  }
}

// Point count a container announces: the sum over its frames of the finest of the three k (the decoder keeps at most
// that many rows per frame).  Host-only parse of the header and the frame slots; nothing is validated beyond their bounds.
extern "C" int pcc_container_points(const uint8_t* h_in, int64_t len, int64_t* h_n_points, int32_t* h_n_frames) {
  PCC_REQUIRE(h_in && len >= 36 && h_n_points, PCC_E_STREAM, "pcc_container_points: container shorter than its header");
  Reader r{h_in, len};
  const int32_t word0 = r.be32();
  const int32_t n_frames = (int32_t)((uint32_t)word0 & 0x00FFFFFFu);
  (void)r.be_f64();
  (void)r.be_f64();
  (void)r.be32();
  (void)r.be32();
  const int32_t ylen = r.be32(), zlen = r.be32();
  (void)r.bytes(ylen);
  (void)r.bytes(zlen);
  PCC_REQUIRE(!r.bad && n_frames >= 0 && n_frames <= 65535, PCC_E_STREAM, "pcc_container_points: truncated container");
  int64_t total = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int32_t pl = r.be32();
    int32_t k[3];
    for (int s = 0; s < 3; ++s) k[s] = r.be32();
    (void)r.bytes(pl);
    PCC_REQUIRE(!r.bad && k[2] >= 0, PCC_E_STREAM, "pcc_container_points: truncated container");
    total += k[2];
  }
  *h_n_points = total;
  if (h_n_frames) *h_n_frames = n_frames;
  return PCC_OK;
}

// pack_batches (codec_parallel.py:474-502) on the device: xyz without the batch column, colours NaN -> 0 and
// clip(c * 255, 0, 255) / 255 in single IEEE operations (the bits numpy's float32 expression gives)
__global__ __launch_bounds__(256) void k_pack_cloud(const int4* __restrict__ coords, const float* __restrict__ colors,
                                                    int64_t n, int32_t* __restrict__ xyz, float* __restrict__ rgb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  xyz[3 * i] = c.y;
  xyz[3 * i + 1] = c.z;
  xyz[3 * i + 2] = c.w;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = colors[3 * i + k];
    v = (v != v) ? 0.0f : v;
    rgb[3 * i + k] = __fdiv_rn(fminf(fmaxf(__fmul_rn(v, 255.0f), 0.0f), 255.0f), 255.0f);
  }
}

extern "C" int pcc_decode_fetch_packed(pcc_codec* cd, int32_t* points, float* colors) {
  PCC_REQUIRE(cd && cd->ctx, PCC_E_ARG, "pcc_decode_fetch_packed: null codec");
  if (cd->rec_n == 0) return PCC_OK;
  PCC_REQUIRE(cd->rec_coords && cd->rec_colors && points && colors, PCC_E_ARG,
              "pcc_decode_fetch_packed: no decoded GOP on this codec, or null destination");
  hipStream_t st = cd->ctx->stream;
  const int64_t n = cd->rec_n;
  CODEC_ALLOC(xyz, int32_t, 3 * n);
  CODEC_ALLOC(rgb, float, 3 * n);
  hipLaunchKernelGGL(k_pack_cloud, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int4*)cd->rec_coords,
                     (const float*)cd->rec_colors, n, xyz, rgb);
  PCC_CHECK_LAUNCH();
  // destinations may be device or host memory (pageable host memory is staged by the runtime)
  PCC_HIP(hipMemcpyAsync(points, xyz, (size_t)n * 12, hipMemcpyDefault, st));
  PCC_HIP(hipMemcpyAsync(colors, rgb, (size_t)n * 12, hipMemcpyDefault, st));
  PCC_HIP(hipStreamSynchronize(st));
  return PCC_OK;
}

extern "C" int pcc_decode_fetch(pcc_codec* cd, int32_t* d_coords, float* d_colors) {
  PCC_REQUIRE(cd && cd->ctx, PCC_E_ARG, "pcc_decode_fetch: null codec");
  if (cd->rec_n == 0) return PCC_OK;
  PCC_REQUIRE(cd->rec_coords && cd->rec_colors, PCC_E_ARG, "pcc_decode_fetch: no decoded GOP on this codec");
  hipStream_t st = cd->ctx->stream;
  if (d_coords) PCC_HIP(hipMemcpyAsync(d_coords, cd->rec_coords, (size_t)cd->rec_n * 16, hipMemcpyDeviceToDevice, st));
  if (d_colors) PCC_HIP(hipMemcpyAsync(d_colors, cd->rec_colors, (size_t)cd->rec_n * 12, hipMemcpyDeviceToDevice, st));
  PCC_HIP(hipStreamSynchronize(st));
  return PCC_OK;
}

// ---------------------------------------------------------------------------- geometry slot, one call each
// (pcc_octree_encode and the forms that take a context: octree2.hip)
extern "C" int pcc_octree_decode(const uint8_t* h_in, int64_t len, int32_t* h_points, int64_t cap_points,
                                 int64_t* h_n_points) {
  int64_t n = 0;
  int depth = 0;
  int32_t org[3];
  PCC_TRY(pcc_octree_peek(h_in, len, &n, &depth, org));
  if (h_n_points) *h_n_points = n;
  if (!h_points || n == 0) return PCC_OK;
  PCC_REQUIRE(cap_points >= n, PCC_E_NOMEM, "pcc_octree_decode: %lld points, capacity %lld", (long long)n,
              (long long)cap_points);
  return pcc_octree_unpack(h_in, len, h_points, cap_points);
}
