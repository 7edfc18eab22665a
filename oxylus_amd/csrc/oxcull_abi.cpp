// oxcull_abi.cpp -- host side of liboxcull.so: the C ABI declared in include/oxcull.h.
//
// Mirrors what RendererInstance::generate_hiz / ::cull_geometry do on the host in the reference
// (Oxylus/src/Render/Passes/CullGeometry.cpp): pick the pipeline variant from the context flags,
// hand out freshly initialised counter buffers, and enqueue the passes in order.  Where the
// reference records vuk passes that run at end_frame, this enqueues HIP kernels on the caller's
// stream; nothing here synchronises except scratch growth and oxc_read_counters.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "oxcull.h"
#include "oxcull_debug.h"
#include "oxcull_kernels.hpp"
#include "oxcull_types.hpp"

using namespace oxc;

namespace {
constexpr uint32_t kSlots = 8192;  // counter-slot ring (1 MiB): a slot is reused kSlots / 2 calls (or seeds) later -- include/oxcull.h states that lifetime
constexpr uint32_t kMaxPackedInstances = 1u << 24;  // MESHLET_INSTANCE_ID_BITS, visbuffer.slang:9-10

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
// (Sizing persistent grids to the exact SGPR-limited residency and balancing chunks per block was tried:
// no gain for the plain kernel, 183 -> 219 us for the HiZ variant; oversubscribed grids schedule better.)
inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
}  // namespace

struct oxc_ctx {
  int device = 0;
  uint32_t num_cus = 256;
  std::string last_error;
  // scratch: one lane per batch element (lane 0 serves the single-frame entry points)
  struct Lane {
    void* arena = nullptr;
    uint64_t arena_bytes = 0;
    uint32_t cap_mesh_instances = 0, cap_meshlets = 0;
    InstCache* cache = nullptr;
    InstCache* cache_alt = nullptr;   // second set of rows: calls alternate, so that a triangle stage still in flight on the side stream keeps the rows of ITS call
    InstCache* view_cache = nullptr;  // [views][M] rows for use_hpb
    uint32_t cap_views = 0;
    uint32_t* mesh_counts = nullptr;
    uint32_t* mesh_offsets = nullptr;
    uint64_t* bits = nullptr;
    uint32_t* m_chunk_counts = nullptr;
    uint32_t* m_supers = nullptr;
    uint32_t* m_tickets = nullptr;
    uint32_t* m_supers_late = nullptr;   // share_pass_tests: the accumulators of the late call, zeroed by the early call's prepare kernel
    uint32_t* m_tickets_late = nullptr;
    uint64_t* camera_test_bits = nullptr;  // share_pass_tests: the early call's "passed frustum and cone" ballots ...
    uint2* step_info = nullptr;        // ... and each wave step's run of mask bits, for the late call of the same frame
    uint64_t* tri_masks = nullptr;
    uint32_t* t_chunk_counts = nullptr;
    uint32_t* t_supers = nullptr;
    uint32_t* t_supers_alt = nullptr;  // alternates with t_supers like the cache rows (prepare zeroes the next call's accumulators on the caller's stream)
  };
  Lane lane[kMaxBatch];
  BatchElem* batch_dev = nullptr;  // device copy of the argument blocks of the current batched call (kMaxBatch elements)
  void* mv_arena = nullptr;        // multi-view meshlet stage (batched views of one scene): view table, groups, step list, per-view chunk arrays
  uint64_t mv_arena_bytes = 0;
  // oxc_build_meshlet_bounds: per-meshlet {min xyz, max xyz} for all meshlets, then per chunk of kBoundsChunk
  // meshlets the compacted triangle normals (768 B each) and their counts
  float* bounds_scratch = nullptr;
  uint32_t bounds_scratch_cap = 0;
  void* raster_scratch = nullptr;  // oxc_draw_visbuffer: list of large triangles + its counter
  uint32_t raster_capacity = 0;    // entries of the big / clip lists (the tile list has twice as many)
  uint32_t raster_capacity_request = 0;  // oxc_debug_set_tuning(OXC_TUNE_RASTER_BIG_CAPACITY): used by the first draw instead of the default
  void* raster_rows = nullptr;     // oxc_draw_visbuffer: one DrawRow per mesh instance
  uint32_t raster_rows_cap = 0;
  void* comm = nullptr;            // ncclComm_t (oxc_comm_init)
  uint32_t comm_rank = 0, comm_world = 0;
  // counter slots
  uint32_t* slots = nullptr;
  uint32_t slot_cursor = 0;
  uint32_t seed_cursor = 0;
  uint32_t* sink = nullptr;
  std::vector<uint32_t> seeded_total = std::vector<uint32_t>(kSlots, 0u);  // per counter slot: list length given to oxc_seed_meshlet_instances (0: unknown)
  // The context owns ONE set of scratch buffers, so its calls are ordered: a call on another stream than the previous
  // call's first waits for that call on the device (order_stream).
  hipStream_t last_stream = nullptr;
  bool has_last_stream = false;
  hipEvent_t order_event = nullptr;
  // async_triangles: the triangle stage of a call runs on `side`, forked from the caller's stream after the meshlet emit.
  // tri[k % kTriRing] describes the stage of lane-0 call number k (call_seq) while it may still be in flight.
  hipStream_t side = nullptr;
  hipEvent_t fork_event = nullptr;
  hipStream_t mv_side = nullptr;
  hipEvent_t mv_fork = nullptr, mv_join = nullptr;  // multi-view batch: the MeshletInstance expansion on `side` beside the meshlet stage
  uint32_t tri_loads = 0;                            // OXC_TUNE_TRI_LOADS: 0 = by the scene (shared geometry -> plain loads), 1 = always `nt`, 2 = always plain
  uint32_t mv_expand_async = 4;                      // blocks per CU the side-stream expansion takes; oxc_debug_set_tuning(OXC_TUNE_MV_EXPAND_ASYNC, 0): in order on the caller's stream (A/B aid)
  struct TriPending {
    hipEvent_t done = nullptr;
    bool valid = false;
    bool late = false;
    const uint32_t* vis = nullptr;  // the sequence's visibility counters (the late list starts at vis[1])
    const void* visible = nullptr;  // visible_meshlet_instances_indices_buffer it reads
    unsigned long long capture_id = 0;  // the HIP-graph capture `done` was recorded in (0: none).  An event recorded outside a capture cannot be
                                        // waited for inside one and the reverse (hipErrorStreamCaptureIsolation): wait_triangles only waits within
                                        // the same capture
  };
  static constexpr uint32_t kTriRing = 4;
  TriPending tri[kTriRing];
  // share_pass_tests: what the last flagged early HiZ call tested, i.e. what lane[0].camera_test_bits / step_info describe
  struct SharedTests {
    bool valid = false;
    uint32_t N = 0, n_host = 0, M = 0, flags = 0, mask_bits = 0;
    const void *meshlet_instances = nullptr, *mask = nullptr, *meshes = nullptr, *transforms = nullptr, *mesh_instances = nullptr, *vis = nullptr;
    oxc_cull_camera camera = {};
    // the early call also did the late call's prepare work (PrepareArgs::slot_late): valid for lane-0 call number `armed_for_call` only
    uint64_t armed_for_call = ~0ull;
    unsigned long long capture_id = 0;  // the capture the arming early call was part of (0: none): an armed late call must be part of the same
                                        // one -- replayed alone, nobody would zero its accumulators again
    uint32_t* late_slot = nullptr;
    uint32_t* late_t_supers = nullptr;
    InstCache* rows = nullptr;
    bool same_inputs(const SharedTests& o) const {
      return N == o.N && n_host == o.n_host && M == o.M && flags == o.flags && mask_bits == o.mask_bits && meshlet_instances == o.meshlet_instances && mask == o.mask && meshes == o.meshes &&
             transforms == o.transforms && mesh_instances == o.mesh_instances && vis == o.vis && std::memcmp(&camera, &o.camera, sizeof camera) == 0;
    }
  };
  SharedTests shared;
  uint32_t last_share_mode = 0;  // oxc_debug_shared_tests_mode
  uint32_t last_tri_loads = 0;   // oxc_debug_tri_loads_mode: 0 = the last call ran no triangle stage, 1 = nt loads, 2 = plain loads
  uint64_t call_seq = 0;  // lane-0 oxc_cull_geometry calls so far: parity selects cache / t_supers
  // resident blocks per CU of the persistent kernels while the two stages share the machine (0 = no limit); oxc_debug_set_tuning
  // overrides the defaults (tuning aid of tools/kbench.py)
  uint32_t async_mtest_per_cu = kAsyncMeshletBlocksPerCU, async_tri_per_cu = kAsyncTriangleBlocksPerCU;
  uint32_t tri_blocks_per_cu = kTriangleBlocksPerCU;  // grid cap of the triangle kernels (blocks walk their chunks with a grid stride)
  uint32_t* dbg_occlusion = nullptr;  // oxc_debug_count_occlusion_candidates: 256 strided counters the counting instantiations of the HiZ meshlet tests add to
  // profiling (oxc_profile_begin/end)
  bool profiling = false;
  struct Rec {
    int id;
    hipEvent_t a, b;
  };
  std::vector<Rec> recs;
};

namespace {

oxc_status fail(oxc_ctx* ctx, oxc_status st, const char* what, hipError_t e = hipSuccess) {
  if (ctx) {
    ctx->last_error = what;
    if (e != hipSuccess) {
      ctx->last_error += ": ";
      ctx->last_error += hipGetErrorString(e);
    }
  }
  return st;
}

#define OXC_ORDER(ctx, stream)                                            \
  do {                                                                    \
    oxc_status _o = order_stream(ctx, static_cast<hipStream_t>(stream)); \
    if (_o != OXC_OK) return _o;                                          \
  } while (0)

#define OXC_HIP(ctx, expr)                                        \
  do {                                                            \
    hipError_t _e = (expr);                                       \
    if (_e != hipSuccess) return fail(ctx, OXC_HIP_ERROR, #expr, _e); \
  } while (0)

bool stream_is_capturing(hipStream_t s);
unsigned long long stream_capture_id(hipStream_t s);

// Device-side ordering of a context's calls across streams (see oxc_ctx::last_stream).  Not across a capture boundary: an event
// recorded outside a capture cannot be waited for inside it (and the reverse), so when `s` is being captured the wait for a
// previous call on ANOTHER stream is skipped -- include/oxcull.h asks the caller to have that stream's work complete (or inside the
// same capture through the caller's own events) before the capture begins.
oxc_status order_stream(oxc_ctx* ctx, hipStream_t s) {
  // (a previous stream that is itself being captured: recording on it would add a node to THAT capture and the wait below would pull `s`
  // into it -- the two are not ordered here either; same promise of the caller as above)
  if (ctx->has_last_stream && ctx->last_stream != s && !stream_is_capturing(s) && !stream_is_capturing(ctx->last_stream)) {
    if (!ctx->order_event) OXC_HIP(ctx, hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
    // (the previous stream may have been destroyed by its owner since -- its work is complete then, nothing to wait for: the record
    // fails, and the error is dropped)
    if (hipEventRecord(ctx->order_event, ctx->last_stream) == hipSuccess)
      OXC_HIP(ctx, hipStreamWaitEvent(s, ctx->order_event, 0));
    else
      (void)hipGetLastError();
  }
  ctx->last_stream = s;
  ctx->has_last_stream = true;
  return OXC_OK;
}

// 0 when `s` is not being captured, else the id of the capture it belongs to
unsigned long long stream_capture_id(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(s, &st, &id) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return st == hipStreamCaptureStatusNone ? 0ull : (id ? id : ~0ull);
}

bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

// Grows the lane's scratch arena when the call needs more than it holds: device sync + free + malloc (documented in
// include/oxcull.h; oxc_reserve up front avoids it).  Never during stream capture: `s` = the calling stream.
oxc_status ensure_capacity(oxc_ctx* ctx, uint32_t mesh_instances, uint32_t meshlets, uint32_t views = 0, uint32_t lane_index = 0, hipStream_t s = nullptr) {
  oxc_ctx::Lane* L = &ctx->lane[lane_index];
  if (mesh_instances <= L->cap_mesh_instances && meshlets <= L->cap_meshlets && views <= L->cap_views && L->arena) return OXC_OK;
  if (stream_is_capturing(s))
    return fail(ctx, OXC_INVALID_ARG, "scratch memory must grow but the stream is being captured: call oxc_reserve (or one un-captured call of this size) first");
  const uint32_t Vw = std::max(views, L->cap_views);
  uint32_t M = std::max(std::max(mesh_instances, L->cap_mesh_instances), 1u);
  uint32_t N = std::max(std::max(meshlets, L->cap_meshlets), 1u);
  const uint32_t m_chunks = cdiv(N, 64u), t_chunks = cdiv(N, kTriChunk);  // meshlet counts: per wave step, >= 64 meshlets
  uint64_t off = 0;
  auto carve = [&](uint64_t bytes) {
    uint64_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const uint64_t o_cache = carve((uint64_t)M * sizeof(InstCache));
  const uint64_t o_cache_alt = carve((uint64_t)M * sizeof(InstCache));
  const uint64_t o_vcache = carve((uint64_t)M * Vw * sizeof(InstCache) + 64);
  const uint64_t o_counts = carve((uint64_t)M * 4);
  const uint64_t o_offsets = carve((uint64_t)M * 4);
  const uint64_t o_bits = carve((uint64_t)cdiv(N, 64) * 8);
  const uint64_t o_mcc = carve((uint64_t)m_chunks * 4);
  const uint64_t o_msup = carve((uint64_t)cdiv(m_chunks, kChunksPerSuper) * 4 * kSuperStride);
  const uint64_t o_mtick = carve((uint64_t)kTicketCounters * 4 * kSuperStride);
  const uint64_t o_msup_late = carve((uint64_t)cdiv(m_chunks, kChunksPerSuper) * 4 * kSuperStride);
  const uint64_t o_mtick_late = carve((uint64_t)kTicketCounters * 4 * kSuperStride);
  const uint64_t o_fb = carve((uint64_t)cdiv(N, 64) * 8);
  const uint64_t o_si = carve((uint64_t)m_chunks * 8);
  const uint64_t o_tm = carve((uint64_t)N * 16);  // one 64-bit pass mask per visible meshlet (two in wide mode)
  const uint64_t o_tcc = carve((uint64_t)t_chunks * 4);
  const uint64_t o_tsup = carve((uint64_t)cdiv(t_chunks, kChunksPerSuper) * 4 * kSuperStride);
  const uint64_t o_tsup_alt = carve((uint64_t)cdiv(t_chunks, kChunksPerSuper) * 4 * kSuperStride);
  OXC_HIP(ctx, hipDeviceSynchronize());  // in-flight work (the side stream's included) may still use the old arena
  for (auto& tp : ctx->tri) tp.valid = false;
  if (lane_index == 0) ctx->shared.valid = false;  // the early call's bits go with the arena
  if (L->arena) OXC_HIP(ctx, hipFree(L->arena));
  L->arena = nullptr;
  hipError_t e = hipMalloc(&L->arena, off);
  if (e != hipSuccess) {
    L->cap_mesh_instances = L->cap_meshlets = 0;
    return fail(ctx, OXC_OUT_OF_MEMORY, "hipMalloc(scratch arena)", e);
  }
  L->arena_bytes = off;
  char* b = static_cast<char*>(L->arena);
  L->cache = reinterpret_cast<InstCache*>(b + o_cache);
  L->cache_alt = reinterpret_cast<InstCache*>(b + o_cache_alt);
  L->view_cache = reinterpret_cast<InstCache*>(b + o_vcache);
  L->cap_views = Vw;
  L->mesh_counts = reinterpret_cast<uint32_t*>(b + o_counts);
  L->mesh_offsets = reinterpret_cast<uint32_t*>(b + o_offsets);
  L->bits = reinterpret_cast<uint64_t*>(b + o_bits);
  L->m_chunk_counts = reinterpret_cast<uint32_t*>(b + o_mcc);
  L->m_supers = reinterpret_cast<uint32_t*>(b + o_msup);
  L->m_tickets = reinterpret_cast<uint32_t*>(b + o_mtick);
  L->m_supers_late = reinterpret_cast<uint32_t*>(b + o_msup_late);
  L->m_tickets_late = reinterpret_cast<uint32_t*>(b + o_mtick_late);
  L->camera_test_bits = reinterpret_cast<uint64_t*>(b + o_fb);
  L->step_info = reinterpret_cast<uint2*>(b + o_si);
  L->tri_masks = reinterpret_cast<uint64_t*>(b + o_tm);
  L->t_chunk_counts = reinterpret_cast<uint32_t*>(b + o_tcc);
  L->t_supers = reinterpret_cast<uint32_t*>(b + o_tsup);
  L->t_supers_alt = reinterpret_cast<uint32_t*>(b + o_tsup_alt);
  L->cap_mesh_instances = M;
  L->cap_meshlets = N;
  return OXC_OK;
}

// Per-call slots come from the lower half of the ring, seed slots (oxc_seed_meshlet_instances: they
// must outlive many calls) from the upper half, so a wrapping call ring never lands on a live seed.
uint32_t* next_slot(oxc_ctx* ctx) {
  uint32_t* s = ctx->slots + (size_t)(ctx->slot_cursor % (kSlots / 2)) * SLOT_U32S;
  ctx->slot_cursor++;
  return s;
}
uint32_t* next_seed_slot(oxc_ctx* ctx) {
  uint32_t* s = ctx->slots + (size_t)(kSlots / 2 + ctx->seed_cursor % (kSlots / 2)) * SLOT_U32S;
  ctx->seed_cursor++;
  return s;
}

// ---- async_triangles: what is in flight on the context's own stream ----
// Makes `s` wait for every pending triangle stage for which `needed` says so.
// Only within one capture (or outside any): a stage recorded outside a capture cannot be waited for inside it -- include/oxcull.h asks the
// caller to have joined before the capture begins -- and a stage recorded INSIDE a capture that has ended is ordered by the graph itself at
// every replay (the capture could not end before oxc_join_triangles), so from un-captured calls it is simply forgotten.
// retire: the caller is an ordered call of the context on `s` (OXC_ORDER ran): everything the context does later is behind this wait, so
// the entry need not be waited for again.
template <class Pred>
oxc_status wait_triangles(oxc_ctx* ctx, hipStream_t s, Pred needed, bool retire = false) {
  const unsigned long long cid = stream_capture_id(s);
  for (auto& tp : ctx->tri) {
    if (!tp.valid) continue;
    if (tp.capture_id != cid) {
      // An entry of another capture is forgotten by an un-captured call only once that capture has ENDED (the graph orders the stage at
      // every replay).  While it is still open -- the side stream joined it through the fork event and stays in capture mode until the
      // stage is joined -- the entry must survive an un-captured call made in between: the oxc_join_triangles inside the capture still
      // has to find it (round-4 advisor finding).
      if (cid == 0 && !(ctx->side && stream_capture_id(ctx->side) == tp.capture_id)) tp.valid = false;
      continue;
    }
    if (!needed(tp)) continue;
    OXC_HIP(ctx, hipStreamWaitEvent(s, tp.done, 0));
    if (retire) tp.valid = false;
  }
  return OXC_OK;
}
oxc_status join_triangles(oxc_ctx* ctx, hipStream_t s) {
  return wait_triangles(ctx, s, [](const oxc_ctx::TriPending&) { return true; }, true);
}
#define OXC_JOIN(ctx, stream)                                               \
  do {                                                                      \
    oxc_status _j = join_triangles(ctx, static_cast<hipStream_t>(stream));  \
    if (_j != OXC_OK) return _j;                                            \
  } while (0)

// Brackets one kernel launch with events while profiling is on.
struct KernelTimer {
  oxc_ctx* ctx;
  hipStream_t s;
  oxc_ctx::Rec r;
  bool on;
  KernelTimer(oxc_ctx* c, int id, hipStream_t st) : ctx(c), s(st), on(c->profiling) {
    if (!on) return;
    r.id = id;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(r.a, s);
  }
  ~KernelTimer() {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    ctx->recs.push_back(r);
  }
};

bool image_ok(const oxc_image& im) { return im.dptr && im.width && im.height && im.levels >= 1 && im.levels <= 13; }

}  // namespace

// Everything oxc_cull_geometry derives from (frame, context) before it touches the device.
struct CallInfo {
  uint32_t stages, M, N, views;
  bool do_meshes, do_meshlets, do_tris, occl, late;
};

static oxc_status check_call(oxc_ctx* ctx, const oxc_prepared_frame* f, const oxc_cull_geometry_context* c, CallInfo& ci) {
  if (!f || !c || c->struct_size != sizeof(oxc_cull_geometry_context)) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: bad frame/context struct");
  const uint32_t stages = c->stages ? c->stages : (uint32_t)OXC_STAGE_ALL;
  const uint32_t M = f->mesh_instance_count, N = f->max_meshlet_instance_count;
  const bool do_meshes = c->init_cull_meshes && (stages & OXC_STAGE_MESHES);
  const bool do_meshlets = (stages & OXC_STAGE_MESHLETS) != 0;
  const bool do_tris = (stages & OXC_STAGE_TRIANGLES) != 0;
  if (c->cull_camera.mesh_instance_count != M) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: cull_camera.mesh_instance_count != frame.mesh_instance_count");
  if (M && (!f->meshes_buffer.dptr || !f->transforms_world_buffer.dptr || !f->mesh_instances_buffer.dptr))
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: meshes/transforms/mesh_instances buffer missing");
  if (f->mesh_instances_buffer.bytes < (uint64_t)M * sizeof(GpuMeshInstance)) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: mesh_instances_buffer too small");
  if (N && (!f->meshlet_instances_buffer.dptr || f->meshlet_instances_buffer.bytes < (uint64_t)N * 8))
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: meshlet_instances_buffer missing or < 8*N bytes");
  if (do_meshlets && N && (!f->visible_meshlet_instances_indices_buffer.dptr || f->visible_meshlet_instances_indices_buffer.bytes < (uint64_t)N * 4))
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: visible_meshlet_instances_indices_buffer missing or < 4*N bytes");
  if (do_tris && N) {
    if (c->wide_triangle_index > 2u) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: wide_triangle_index must be 0, 1 or 2");
    const uint64_t tris_per_meshlet = c->wide_triangle_index ? 128u : 64u, index_bytes = c->wide_triangle_index == 2u ? 8u : 4u;
    if (!f->reordered_indices_buffer.dptr || f->reordered_indices_buffer.bytes < (uint64_t)N * tris_per_meshlet * 3 * index_bytes)
      return fail(ctx, OXC_INVALID_ARG, "cull_geometry: reordered_indices_buffer missing or < N*64*3*4 bytes (N*128*3*4 with wide_triangle_index = 1, N*128*3*8 with 2)");
    // (wide_triangle_index = 2, {id, corner} pairs: no id limit below 2^32 -- N is a u32)
    if (c->wide_triangle_index != 2u && N > (c->wide_triangle_index ? kMaxPackedInstances / 2 : kMaxPackedInstances))
      return fail(ctx, OXC_INVALID_ARG, "cull_geometry: too many meshlet instances for the packed index (2^24, visbuffer.slang:9-14; 2^23 with wide_triangle_index = 1; pairs, wide_triangle_index = 2, have no limit)");
  }
  const bool occl = (c->cull_flags & OXC_CULL_TEST_OCCLUSION) != 0;
  const bool late = (c->cull_flags & OXC_CULL_LATE_PASS) != 0;
  if (c->use_hiz && do_meshlets) {
    if (!image_ok(c->hiz_attachment)) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: use_hiz without a hiz_attachment");
    if (occl && N && (!f->meshlet_instance_visibility_mask_buffer.dptr || f->meshlet_instance_visibility_mask_buffer.bytes < (uint64_t)cdiv(N, 32u) * 4u))
      return fail(ctx, OXC_INVALID_ARG, "cull_geometry: visibility mask buffer missing or < ceil(N/32)*4 bytes");
  }
  if (c->small_triangle_cull > 1u) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: small_triangle_cull must be 0 or 1");
  if (c->share_pass_tests > 1u || c->unordered_output > 1u) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: share_pass_tests must be 0 or 1, unordered_output 0 or 1");
  if (c->implicit_meshlet_instances > 1u || c->_reserved1 != 0u) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: implicit_meshlet_instances must be 0 or 1, _reserved1 0");
  if (c->meshlet_instance_runs_buffer.dptr && c->meshlet_instance_runs_buffer.bytes < (uint64_t)M * 8u)
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: meshlet_instance_runs_buffer < 8 bytes per mesh instance");
  if (c->implicit_meshlet_instances && (!c->meshlet_instance_runs_buffer.dptr || do_tris || !do_meshes))
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: implicit_meshlet_instances needs meshlet_instance_runs_buffer, cull_meshes in the call and no triangle stage");
  const uint32_t views = c->use_hpb ? c->vsm_clipmap_count : 0u;
  if (c->use_hpb && do_meshlets) {
    if (c->use_hiz) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: use_hiz and use_hpb are exclusive (CullGeometry.cpp:129,199)");
    if (views == 0 || views > 16) return fail(ctx, OXC_INVALID_ARG, "cull_geometry: vsm_clipmap_count must be 1..16");
    if (!c->vsm_clipmaps_buffer.dptr || c->vsm_clipmaps_buffer.bytes < (uint64_t)views * sizeof(oxc_virtual_clipmap) || !c->vsm_clipmap_dirty_flags_buffer.dptr ||
        c->vsm_clipmap_dirty_flags_buffer.bytes < (uint64_t)views * 4u)
      return fail(ctx, OXC_INVALID_ARG, "cull_geometry: use_hpb needs vsm_clipmaps_buffer (76 B per clipmap) / vsm_clipmap_dirty_flags_buffer (4 B per clipmap)");
    const oxc_image_array_u8& h = c->hpb_attachment;
    if (!h.dptr || !h.width || !h.height || h.layers < views || h.levels < 1 || h.levels > 13)
      return fail(ctx, OXC_INVALID_ARG, "cull_geometry: use_hpb without a valid hpb_attachment (layers >= clipmaps, 1..13 levels)");
    for (uint32_t k = 0; k < h.levels; k++) {  // the page test addresses the pyramid with 32-bit byte offsets (test_vsm_page)
      const uint64_t mw = std::max(1u, h.width >> k), mh = std::max(1u, h.height >> k);
      if (h.level_offset[k] + (uint64_t)h.layers * mw * mh > 0xFFFFFFFFull)
        return fail(ctx, OXC_INVALID_ARG, "cull_geometry: hpb_attachment larger than 4 GiB (the reference's is 64 x 64 pages x 10 clipmaps, Shadowmaps.cpp:84-98)");
    }
  }
  if (!c->init_cull_meshes && (!c->visibility_buffer.dptr || !c->cull_meshlets_cmd_buffer.dptr))
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: init_cull_meshes=false needs the visibility/cull_meshlets_cmd buffers of the sequence");


  ci.stages = stages;
  ci.M = M;
  ci.N = N;
  ci.views = views;
  ci.do_meshes = do_meshes;
  ci.do_meshlets = do_meshlets;
  ci.do_tris = do_tris;
  ci.occl = occl;
  ci.late = late;
  return OXC_OK;
}

extern "C" {

uint32_t oxc_abi_version(void) { return OXC_ABI_VERSION; }

oxc_status oxc_create(int device, oxc_ctx** out) {
  if (!out) return OXC_INVALID_ARG;
  *out = nullptr;
  oxc_ctx* ctx = new (std::nothrow) oxc_ctx();
  if (!ctx) return OXC_OUT_OF_MEMORY;
  ctx->device = device;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    delete ctx;
    return OXC_HIP_ERROR;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    ctx->num_cus = (uint32_t)prop.multiProcessorCount;
  e = hipMalloc(reinterpret_cast<void**>(&ctx->slots), (size_t)kSlots * SLOT_U32S * 4 + 256);
  if (e != hipSuccess) {
    delete ctx;
    return OXC_OUT_OF_MEMORY;
  }
  // (hipMemset runs on the NULL stream, which does not order against the non-blocking streams callers pass in: without the synchronisation a
  // fill that starts late -- the first one of a process loads its kernel lazily -- lands on slots the first calls have already written)
  (void)hipMemset(ctx->slots, 0, (size_t)kSlots * SLOT_U32S * 4 + 256);
  (void)hipDeviceSynchronize();
  ctx->sink = ctx->slots + (size_t)kSlots * SLOT_U32S;
  *out = ctx;
  return OXC_OK;
}

void oxc_destroy(oxc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& ln : ctx->lane)
    if (ln.arena) (void)hipFree(ln.arena);
  if (ctx->batch_dev) (void)hipFree(ctx->batch_dev);
  if (ctx->mv_arena) (void)hipFree(ctx->mv_arena);
  if (ctx->bounds_scratch) (void)hipFree(ctx->bounds_scratch);
  if (ctx->raster_scratch) (void)hipFree(ctx->raster_scratch);
  if (ctx->raster_rows) (void)hipFree(ctx->raster_rows);
  if (ctx->comm) (void)oxc_comm_destroy(ctx);
  if (ctx->slots) (void)hipFree(ctx->slots);
  if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
  if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
  if (ctx->mv_side) (void)hipStreamDestroy(ctx->mv_side);
  if (ctx->mv_fork) (void)hipEventDestroy(ctx->mv_fork);
  if (ctx->mv_join) (void)hipEventDestroy(ctx->mv_join);
  for (auto& tp : ctx->tri)
    if (tp.done) (void)hipEventDestroy(tp.done);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  delete ctx;
}

const char* oxc_last_error(const oxc_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

oxc_status oxc_reserve(oxc_ctx* ctx, uint32_t max_mesh_instances, uint32_t max_meshlet_instances) {
  if (!ctx) return OXC_INVALID_ARG;
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  return ensure_capacity(ctx, max_mesh_instances, max_meshlet_instances);  // (synchronises the device when it has to grow)
}

oxc_status oxc_generate_hiz(oxc_ctx* ctx, const oxc_main_geometry_context* c, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!c || c->struct_size != sizeof(oxc_main_geometry_context)) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: bad context struct");
  const oxc_image& d = c->depth_attachment;
  const oxc_image& h = c->hiz_attachment;
  if (!d.dptr || !d.width || !d.height) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: depth_attachment missing");
  if (!image_ok(h)) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: hiz_attachment missing or levels not in 1..13");
  if (h.width > 4096 || h.height > 4096) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: hiz extent > 4096 (13 mips) unsupported");
  HizArgs a;
  a.depth = reinterpret_cast<const float*>(static_cast<const char*>(d.dptr) + d.level_offset[0]);
  a.hiz = static_cast<float*>(h.dptr);
  a.dw = d.width;
  a.dh = d.height;
  a.w = h.width;
  a.h = h.height;
  a.levels = std::min(h.levels, 13u);  // CullGeometry.cpp:24
  for (uint32_t k = 0; k < 13; k++) {
    if (k < a.levels && (h.level_offset[k] & 3u)) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: level offset not 4-byte aligned");
    a.level_off[k] = k < a.levels ? (uint32_t)(h.level_offset[k] / 4) : 0u;
  }
  const bool tiled = (a.w % 64 == 0) && (a.h % 64 == 0);
  if (!tiled && (uint64_t)a.w * a.h > 4096) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: extent must be a multiple of 64 or <= 4096 texels");
  if (tiled && (h.level_offset[0] & 15u)) return fail(ctx, OXC_INVALID_ARG, "generate_hiz: mip 0 must be 16-byte aligned");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  {
    KernelTimer t(ctx, OXC_K_HIZ, static_cast<hipStream_t>(hip_stream));
    launch_hiz(a, ctx->num_cus, static_cast<hipStream_t>(hip_stream));
  }
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_seed_meshlet_instances(oxc_ctx* ctx, oxc_cull_geometry_context* c, uint32_t total, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!c || c->struct_size != sizeof(oxc_cull_geometry_context)) return fail(ctx, OXC_INVALID_ARG, "seed: bad context struct");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  uint32_t* slot = next_seed_slot(ctx);
  ctx->shared.valid = false;  // (share_pass_tests: a new list)
  ctx->seeded_total[(slot - ctx->slots) / SLOT_U32S] = total;
  launch_seed_slot(slot, total, static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  c->visibility_buffer = {slot + SLOT_VIS, 12};
  c->cull_meshlets_cmd_buffer = {slot + SLOT_MESHLETS_CMD, 12};
  return OXC_OK;
}

oxc_status oxc_cull_geometry(oxc_ctx* ctx, const oxc_prepared_frame* f, oxc_cull_geometry_context* c, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  CallInfo ci;
  {
    oxc_status vst = check_call(ctx, f, c, ci);
    if (vst != OXC_OK) return vst;
  }
  const uint32_t M = ci.M, N = ci.N, views = ci.views;
  const bool do_meshes = ci.do_meshes, do_meshlets = ci.do_meshlets, do_tris = ci.do_tris, occl = ci.occl, late = ci.late;
  if (c->implicit_meshlet_instances)
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry: implicit_meshlet_instances is accepted by oxc_cull_geometry_batch's multi-view path only (this call's meshlet test reads the records)");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  oxc_status st = ensure_capacity(ctx, M, N, views, 0, s);
  if (st != OXC_OK) return st;
  OXC_ORDER(ctx, hip_stream);
  ctx->last_share_mode = 0;
  ctx->last_tri_loads = 0;
  if (do_meshes) ctx->shared.valid = false;  // (share_pass_tests: the MeshletInstance list is rebuilt; a flagged early call marks its results valid below)

  // ---- async_triangles bookkeeping (include/oxcull.h).  Calls alternate between two sets of instance rows / triangle-count
  // accumulators, so that what this call's prepare kernel writes on `s` is never what a triangle stage still in flight on the
  // side stream reads; the set of this call was last used by the call before the previous one.
  oxc_ctx::Lane& L0 = ctx->lane[0];
  const uint64_t call_no = ctx->call_seq++;
  const bool async = c->async_triangles != 0 && do_tris;
  InstCache* cache = (call_no & 1u) ? L0.cache_alt : L0.cache;
  uint32_t* t_supers = (call_no & 1u) ? L0.t_supers_alt : L0.t_supers;
  oxc_ctx::TriPending& my_tri = ctx->tri[call_no % oxc_ctx::kTriRing];
  my_tri.valid = false;  // (the stage of call_no - kTriRing: every call since has waited for it where it mattered, and the side stream is in order)
  if (!async || do_meshes) {
    // in order on `s`: the triangle stage shares tri_masks / chunk counts with the side stream; cull_meshes rewrites the
    // MeshletInstance records a pending stage reads
    OXC_JOIN(ctx, s);
  } else if (call_no >= 2) {
    const oxc_ctx::TriPending* old = &ctx->tri[(call_no - 2) % oxc_ctx::kTriRing];
    oxc_status wst = wait_triangles(ctx, s, [old](const oxc_ctx::TriPending& tp) { return &tp == old; });
    if (wst != OXC_OK) return wst;
  }
  // Two kernels only run side by side when neither fills every wave slot (tools/overlap_probe.py): with async_triangles the
  // persistent kernels of both stages take a share of the CUs' slots instead of all of them.
  const uint32_t mtest_limit = (c->async_triangles && ctx->async_mtest_per_cu) ? ctx->async_mtest_per_cu * ctx->num_cus : 0u;
  const uint32_t tri_grid_cap = (async && ctx->async_tri_per_cu) ? ctx->async_tri_per_cu * ctx->num_cus : ctx->num_cus * ctx->tri_blocks_per_cu;
  const void* const visible_buf = f->visible_meshlet_instances_indices_buffer.dptr;
  // the meshlet emit of this call overwrites the visible list: wait for the pending stages that still read it -- all but the
  // early triangle stage of the same sequence when this is its late call (the late list starts behind the early one)
  auto wait_for_visible_list = [&](const uint32_t* vis_now) -> oxc_status {
    return wait_triangles(ctx, s, [&](const oxc_ctx::TriPending& tp) { return !(late && !tp.late && tp.vis == vis_now && tp.visible == visible_buf); });
  };

  uint32_t* slot = c->init_cull_meshes ? next_slot(ctx) : nullptr;  // (otherwise taken below: a late call may find its slot prepared)
  uint32_t* vis;
  uint32_t* meshlets_cmd;
  uint32_t n_host = 0;  // list length when the host knows it (oxc_seed_meshlet_instances), else 0
  if (!c->init_cull_meshes) {
    const uint32_t* v = static_cast<const uint32_t*>(c->visibility_buffer.dptr);
    if (v >= ctx->slots && v < ctx->slots + (size_t)kSlots * SLOT_U32S && ((v - ctx->slots) % SLOT_U32S) == SLOT_VIS)
      n_host = ctx->seeded_total[(v - ctx->slots) / SLOT_U32S];
    if (n_host > N) n_host = 0;
  }
  if (c->init_cull_meshes) {
    vis = slot + SLOT_VIS;
    meshlets_cmd = slot + SLOT_MESHLETS_CMD;
    c->visibility_buffer = {vis, 12};
    c->cull_meshlets_cmd_buffer = {meshlets_cmd, 12};
  } else {
    vis = static_cast<uint32_t*>(c->visibility_buffer.dptr);
    meshlets_cmd = static_cast<uint32_t*>(c->cull_meshlets_cmd_buffer.dptr);
  }
  // ---- share_pass_tests (include/oxcull.h): is this the early call that publishes its frustum + cone results (1), or the late call that
  // continues it (2)?  An early call in order on one stream (no async_triangles) also does the late call's prepare work in its own
  // prepare kernel -- accumulators, counter slot; the instance rows are the same -- so that a late call that follows it immediately
  // launches no prepare kernel at all.
  uint32_t share_mode = 0;
  bool armed_late = false, arm_late = false;
  uint32_t* m_supers = L0.m_supers;
  uint32_t* m_tickets = L0.m_tickets;
  const uint32_t mask_bits = (uint32_t)std::min<uint64_t>(f->meshlet_instance_visibility_mask_buffer.bytes / 4u * 32u, 0xFFFFFFFEull);
  // unordered_output (include/oxcull.h): which stages append by themselves
  const bool unord_tris = c->unordered_output != 0u && do_tris;
  const bool unord_meshlets = do_meshlets && !c->use_hpb && !c->use_hiz && c->unordered_output != 0u;  // (the plain meshlet test appends by itself)
  if (c->use_hiz && occl && c->share_pass_tests && do_meshlets && !unord_meshlets) {
    oxc_ctx::SharedTests now;
    now.N = N;
    now.n_host = n_host;
    now.M = M;
    now.flags = c->cull_flags & ~(uint32_t)OXC_CULL_LATE_PASS;
    now.mask_bits = mask_bits;  // (the early call's "one run of mask bits" is a statement about this many bits)
    now.meshlet_instances = f->meshlet_instances_buffer.dptr;
    now.mask = f->meshlet_instance_visibility_mask_buffer.dptr;
    now.meshes = f->meshes_buffer.dptr;
    now.transforms = f->transforms_world_buffer.dptr;
    now.mesh_instances = f->mesh_instances_buffer.dptr;
    now.vis = vis;
    now.camera = c->cull_camera;
    if (!late) {
      share_mode = 1u;
      arm_late = c->async_triangles == 0;
      now.valid = true;
      if (arm_late) {
        now.armed_for_call = call_no + 1;
        now.capture_id = stream_capture_id(s);
        now.late_slot = next_slot(ctx);
        now.late_t_supers = (call_no & 1u) ? L0.t_supers : L0.t_supers_alt;  // the other set: this call's triangle stage uses t_supers
        now.rows = cache;
      }
      ctx->shared = now;
    } else if (!do_meshes && !c->init_cull_meshes && ctx->shared.valid && ctx->shared.same_inputs(now)) {
      share_mode = 2u;
      if (ctx->shared.armed_for_call == call_no && c->async_triangles == 0 && ctx->shared.capture_id == stream_capture_id(s)) {
        armed_late = true;
        slot = ctx->shared.late_slot;
        t_supers = ctx->shared.late_t_supers;
        cache = ctx->shared.rows;
        m_supers = L0.m_supers_late;
        m_tickets = L0.m_tickets_late;
      }
      ctx->shared.valid = false;  // the bits are consumed ONCE: a second late call, or a late call of a later frame whose early call did not
                                  // publish, tests for itself (what the host can check of "nothing was written in between" is little enough)
    }
  } else if (c->use_hiz && do_meshlets && !late) {
    ctx->shared.valid = false;  // an early HiZ call without the flag starts a frame whose late call must not pick up an older frame's bits
  }
  if (!armed_late && ctx->shared.armed_for_call <= call_no) ctx->shared.armed_for_call = ~0ull;  // (armed for this call only)
  if (!slot) slot = next_slot(ctx);
  uint32_t* tri_cmd = slot + SLOT_TRI_CMD;
  uint32_t* draw_cmd = slot + SLOT_DRAW_CMD;
  c->cull_triangles_cmd_buffer = {tri_cmd, 12};
  c->draw_geometry_cmd_buffer = {draw_cmd, 20};

  const uint32_t max_grid = ctx->num_cus * 8;
  const uint32_t m_chunks = cdiv(std::max(N, 1u), kMeshletChunk), t_chunks = cdiv(std::max(N, 1u), kTriChunk);

  // --- prepare (+ cull_meshes test): CullGeometry.cpp:69-117
  PrepareArgs pa;
  pa.meshes = static_cast<const GpuMesh*>(f->meshes_buffer.dptr);
  pa.transforms = static_cast<const float*>(f->transforms_world_buffer.dptr);
  pa.mesh_instances = static_cast<GpuMeshInstance*>(f->mesh_instances_buffer.dptr);
  pa.cache = cache;
  pa.mesh_counts = ctx->lane[0].mesh_counts;
  pa.slot = slot;
  pa.vis = vis;
  pa.meshlets_cmd = meshlets_cmd;
  pa.supers_meshlets = m_supers;
  pa.supers_tris = t_supers;
  pa.tickets = m_tickets;
  pa.slot_late = arm_late ? ctx->shared.late_slot : nullptr;
  pa.supers_meshlets_late = arm_late ? L0.m_supers_late : nullptr;
  pa.supers_tris_late = arm_late ? ctx->shared.late_t_supers : nullptr;
  pa.tickets_late = arm_late ? L0.m_tickets_late : nullptr;
  pa.n_supers_meshlets = cdiv(cdiv(std::max(N, 1u), 64u), kChunksPerSuper);
  pa.n_supers_tris = cdiv(t_chunks, kChunksPerSuper);
  pa.mesh_instance_count = M;
  pa.cull_flags = c->cull_flags;
  pa.do_cull_meshes = do_meshes ? 1u : 0u;
  pa.init_vis = c->init_cull_meshes ? 1u : 0u;
  pa.seed_total = 0;
  pa.cam = c->cull_camera;
  pa.clipmaps = static_cast<const oxc_virtual_clipmap*>(c->vsm_clipmaps_buffer.dptr);
  pa.view_cache = ctx->lane[0].view_cache;
  const uint32_t prep_threads = std::max(std::max(M * 8u, pa.n_supers_tris), 1u);  // 8 lanes per mesh instance
  if (!armed_late) {  // (an armed late call: the early call of the frame did all of this)
    KernelTimer t(ctx, OXC_K_PREPARE, s);
    launch_prepare(pa, std::min(cdiv(prep_threads, 256), max_grid), (c->use_hpb && do_meshlets) ? views : 0u, s);
  }
  if (do_meshes) {
    {
      KernelTimer t(ctx, OXC_K_MESHES_SCAN, s);
      launch_scan_mesh_counts(ctx->lane[0].mesh_counts, ctx->lane[0].mesh_offsets, M, N, vis, meshlets_cmd, s);
    }
    KernelTimer t(ctx, OXC_K_MESHES_EXPAND, s);
    launch_expand(ctx->lane[0].mesh_counts, ctx->lane[0].mesh_offsets, M, N, f->meshlet_instances_buffer.dptr, std::max(std::min(cdiv(M, 4), max_grid), 1u), s);
  }

  // --- meshlet stage: CullGeometry.cpp:129-335
  if (do_meshlets && c->use_hpb) {  // CullGeometry.cpp:199-273
    HpbTestArgs ha;
    std::memset(&ha, 0, sizeof ha);
    ha.cache = cache;
    ha.view_cache = ctx->lane[0].view_cache;
    ha.meshlet_instances = static_cast<const GpuMeshletInstance*>(f->meshlet_instances_buffer.dptr);
    ha.vis = vis;
    ha.bits = ctx->lane[0].bits;
    ha.chunk_counts = ctx->lane[0].m_chunk_counts;
    ha.supers = ctx->lane[0].m_supers;
    ha.tickets = ctx->lane[0].m_tickets;
    ha.clipmaps = static_cast<const oxc_virtual_clipmap*>(c->vsm_clipmaps_buffer.dptr);
    ha.dirty = static_cast<const uint32_t*>(c->vsm_clipmap_dirty_flags_buffer.dptr);
    ha.clipmap_count = views;
    ha.mesh_instance_count = M;
    ha.n_cap = N;
    const oxc_image_array_u8& h = c->hpb_attachment;
    ha.hpb_data = static_cast<const uint8_t*>(h.dptr);
    ha.hpb_w = h.width;
    ha.hpb_h = h.height;
    ha.hpb_layers = h.layers;
    ha.hpb_levels = h.levels;
    for (uint32_t k = 0; k < h.levels && k < 13; k++) ha.hpb_level_off[k] = (uint32_t)h.level_offset[k];
    std::memcpy(ha.light_dir, c->cull_camera.position, 12);  // camera.position = -light_dir (Shadowmaps.cpp:433-437)
    {
      KernelTimer t(ctx, OXC_K_MESHLETS_TEST, s);
      launch_hpb_test(ha, std::min(m_chunks, max_grid), s);
    }
    MeshletEmitArgs ea;
    ea.n_host = 0;
    ea.n_cap = N;
    ea.count_meshlets = 64u * kGroupsPerWave;  // one count per wave step
    ea.bits = ctx->lane[0].bits;
    ea.chunk_counts = ctx->lane[0].m_chunk_counts;
    ea.supers = ctx->lane[0].m_supers;
    ea.vis = vis;
    ea.tri_cmd = tri_cmd;
    ea.out = static_cast<uint32_t*>(f->visible_meshlet_instances_indices_buffer.dptr);
    {
      oxc_status wst = wait_for_visible_list(vis);
      if (wst != OXC_OK) return wst;
    }
    KernelTimer t(ctx, OXC_K_MESHLETS_EMIT, s);
    launch_meshlets_emit(ea, false, false, std::min(cdiv(std::max(N, 1u), kMeshletSpan), max_grid), s);
  } else if (do_meshlets) {
    MeshletTestArgs ta;
    std::memset(&ta, 0, sizeof ta);
    ta.n_host = n_host;
    ta.n_cap = N;
    ta.mask_bits = mask_bits;
    ta.cache = cache;
    ta.meshlet_instances = static_cast<const GpuMeshletInstance*>(f->meshlet_instances_buffer.dptr);
    ta.vis = vis;
    ta.mask = static_cast<uint32_t*>(f->meshlet_instance_visibility_mask_buffer.dptr);
    ta.bits = ctx->lane[0].bits;
    ta.chunk_counts = ctx->lane[0].m_chunk_counts;
    ta.supers = m_supers;
    if (c->use_hiz) {
      ta.tickets = m_tickets;  // the HiZ variants take their work dynamically (step cost varies 10x)
      const oxc_image& h = c->hiz_attachment;
      ta.hiz_data = static_cast<const float*>(h.dptr);
      ta.hiz_w = h.width;
      ta.hiz_h = h.height;
      ta.hiz_levels = h.levels;
      for (uint32_t k = 0; k < h.levels && k < 13; k++) ta.hiz_level_off[k] = (uint32_t)(h.level_offset[k] / 4);
      // LDS-staged top of the pyramid: the longest suffix of levels that fits kHizLdsTexels floats
      uint32_t first = h.levels, used = 0;
      while (first > 0) {
        const uint64_t n = (uint64_t)std::max(1u, h.width >> (first - 1)) * std::max(1u, h.height >> (first - 1));
        if (used + n > kHizLdsTexels) break;
        used += (uint32_t)n;
        first--;
      }
      ta.hiz_lds_first = first;
      uint32_t off = 0;
      for (uint32_t k = first; k < h.levels; k++) {
        ta.hiz_lds_off[k] = off;
        off += std::max(1u, h.width >> k) * std::max(1u, h.height >> k);
      }
    }
    ta.near_clip = c->cull_camera.near_clip;
    std::memcpy(ta.cam_pos, c->cull_camera.position, 12);
    ta.dbg_occlusion = ctx->dbg_occlusion;  // (oxc_debug_count_occlusion_candidates; null unless a measurement asked for it)
    if (share_mode) {  // include/oxcull.h: the late call of a frame reuses the early call's frustum + cone results (decided above)
      ta.share = share_mode;
      ta.camera_test_bits = ctx->lane[0].camera_test_bits;
      ta.step_info = ctx->lane[0].step_info;
    }
    ctx->last_share_mode = armed_late ? 3u : ta.share;
    if (unord_meshlets) {  // the test kernel appends: it writes the visible list (wait for the stages that still read it) and both counters
      ta.out = static_cast<uint32_t*>(f->visible_meshlet_instances_indices_buffer.dptr);
      ta.count_a = tri_cmd;
      oxc_status wst = wait_for_visible_list(vis);
      if (wst != OXC_OK) return wst;
      KernelTimer t(ctx, late ? OXC_K_MESHLETS_TEST_LATE : OXC_K_MESHLETS_TEST, s);
      launch_meshlets_test(ta, c->use_hiz != 0, occl, late, std::min(m_chunks, max_grid), ctx->num_cus, mtest_limit, s);
    } else {
      {
        KernelTimer t(ctx, late ? OXC_K_MESHLETS_TEST_LATE : OXC_K_MESHLETS_TEST, s);
        launch_meshlets_test(ta, c->use_hiz != 0, occl, late, std::min(m_chunks, max_grid), ctx->num_cus, mtest_limit, s);
      }
      MeshletEmitArgs ea;
      ea.n_host = n_host;
      ea.n_cap = N;
      ea.count_meshlets = c->use_hiz ? 64u * kHizGroupsPerWave : 64u * kPlainGroups;  // one count per wave step: 64 * groups per wave
      ea.bits = ctx->lane[0].bits;
      ea.chunk_counts = ctx->lane[0].m_chunk_counts;
      ea.supers = m_supers;
      ea.vis = vis;
      ea.tri_cmd = tri_cmd;
      ea.out = static_cast<uint32_t*>(f->visible_meshlet_instances_indices_buffer.dptr);
      {
        oxc_status wst = wait_for_visible_list(vis);
        if (wst != OXC_OK) return wst;
      }
      KernelTimer t(ctx, late ? OXC_K_MESHLETS_EMIT_LATE : OXC_K_MESHLETS_EMIT, s);
      launch_meshlets_emit(ea, c->use_hiz != 0, late, std::min(cdiv(std::max(N, 1u), kMeshletSpan), max_grid), s);
    }
  }

  // --- triangle stage: CullGeometry.cpp:337-403
  if (do_tris) {
    hipStream_t ts = s;
    if (async) {  // fork: everything enqueued on `s` so far (this call's meshlet emit, the caller's earlier consumers of the index list) precedes the stage
      if (!ctx->side) OXC_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
      if (!ctx->fork_event) OXC_HIP(ctx, hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
      OXC_HIP(ctx, hipEventRecord(ctx->fork_event, s));
      OXC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->fork_event, 0));
      ts = ctx->side;
    }
    TriTestArgs tt;
    tt.cache = cache;
    tt.meshlet_instances = static_cast<const GpuMeshletInstance*>(f->meshlet_instances_buffer.dptr);
    tt.visible = static_cast<const uint32_t*>(f->visible_meshlet_instances_indices_buffer.dptr);
    tt.vis = vis;
    tt.tri_cmd = tri_cmd;
    tt.tri_masks = ctx->lane[0].tri_masks;
    tt.chunk_counts = ctx->lane[0].t_chunk_counts;
    tt.supers = t_supers;
    tt.resolution[0] = c->cull_camera.resolution[0];
    tt.resolution[1] = c->cull_camera.resolution[1];
    tt.draw_cmd = draw_cmd;
    tt.out = static_cast<uint32_t*>(f->reordered_indices_buffer.dptr);
    tt.ticket = t_supers;  // (zeroed by this call's prepare kernel; the fused form has no other use for the accumulators)
    tt.ticket_count = std::max(pa.n_supers_tris, 1u);
    // Geometry shared between instances (the engine's case: a few Mesh records, many MeshInstances) is read with plain loads -- the next visible
    // meshlet of another instance finds the lines in L2; unique geometry is streamed once with `nt` loads (oxcull_kernels.hip, OXC_TRI_LOAD_*).
    // "Shared" = at least four mesh instances per Mesh record of the caller's meshes_buffer.
    const uint64_t mesh_records = f->meshes_buffer.bytes / sizeof(GpuMesh);
    const bool cached_loads = ctx->tri_loads == 2u || (ctx->tri_loads == 0u && mesh_records != 0u && (uint64_t)M >= 4u * mesh_records);
    ctx->last_tri_loads = (cached_loads && !c->small_triangle_cull) ? 2u : 1u;
    if (unord_tris) {  // one launch: test + expansion per work item (a span of kFusedTriSpan visible meshlets, or a chunk of it)
      KernelTimer t(ctx, late ? OXC_K_TRIANGLES_TEST_LATE : OXC_K_TRIANGLES_TEST, ts);
      // (the default cap is "one resident round": of the instantiation that runs, which the launcher knows; a cap set by hand stands)
      const bool default_cap = !(async && ctx->async_tri_per_cu) && ctx->tri_blocks_per_cu == kTriangleBlocksPerCU;
      launch_tris_fused(tt, late, c->wide_triangle_index, c->small_triangle_cull != 0, cached_loads, std::min(cdiv(std::max(N, 1u), kFusedTriSpan), tri_grid_cap),
                        default_cap ? ctx->num_cus : 0u, ts);
    } else {
    {
      KernelTimer t(ctx, late ? OXC_K_TRIANGLES_TEST_LATE : OXC_K_TRIANGLES_TEST, ts);
      // (the ordered test kernel walks 64-meshlet chunks with a grid stride: with the default cap it takes four resident rounds of blocks and
      //  lets the dispatcher balance them -- late launch 149.6 -> 140.3 us on the configs[2] frame; a cap set by hand stands)
      const bool default_cap = !(async && ctx->async_tri_per_cu) && ctx->tri_blocks_per_cu == kTriangleBlocksPerCU;
      launch_tris_test(tt, late, c->wide_triangle_index != 0, c->small_triangle_cull != 0, cached_loads, std::min(t_chunks, (default_cap && !c->wide_triangle_index) ? ctx->num_cus * 32u : tri_grid_cap), ts);  // (WIDE, 4 waves per SIMD: 8 per CU measured better than 32)
    }
    TriEmitArgs te;
    te.tri_masks = ctx->lane[0].tri_masks;
    te.visible = tt.visible;
    te.vis = vis;
    te.tri_cmd = tri_cmd;
    te.chunk_counts = ctx->lane[0].t_chunk_counts;
    te.supers = t_supers;
    te.draw_cmd = draw_cmd;
    te.out = static_cast<uint32_t*>(f->reordered_indices_buffer.dptr);
    {
      KernelTimer t(ctx, late ? OXC_K_TRIANGLES_EMIT_LATE : OXC_K_TRIANGLES_EMIT, ts);
      launch_tris_emit(te, late, c->wide_triangle_index, std::min(cdiv(std::max(N, 1u), kTriSpan), tri_grid_cap), ts);
    }
    }
    if (async) {
      if (!my_tri.done) OXC_HIP(ctx, hipEventCreateWithFlags(&my_tri.done, hipEventDisableTiming));
      OXC_HIP(ctx, hipEventRecord(my_tri.done, ts));
      my_tri.valid = true;
      my_tri.late = late;
      my_tri.vis = vis;
      my_tri.visible = visible_buf;
      my_tri.capture_id = stream_capture_id(s);  // (the side stream joined the caller's capture through the fork event)
    }
  }
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_join_triangles(oxc_ctx* ctx, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);  // an ordered call of the context like any other: what it joined stays joined for the calls that follow
  return join_triangles(ctx, static_cast<hipStream_t>(hip_stream));
}

oxc_status oxc_cull_geometry_batch(oxc_ctx* ctx, uint32_t count, const oxc_prepared_frame* frames, oxc_cull_geometry_context* contexts,
                                   void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (count == 0) return OXC_OK;
  if (!frames || !contexts) return fail(ctx, OXC_INVALID_ARG, "cull_geometry_batch: null frames/contexts");
  CallInfo ci[kMaxBatch];
  bool fusable = count > 1 && count <= kMaxBatch;
  for (uint32_t e = 0; fusable && e < count; e++) {
    oxc_status vst = check_call(ctx, &frames[e], &contexts[e], ci[e]);
    if (vst != OXC_OK) return vst;
    const oxc_cull_geometry_context& c = contexts[e];
    fusable = !c.use_hiz && !c.use_hpb && !c.wide_triangle_index && !c.small_triangle_cull && !ci[e].late && ci[e].stages == ci[0].stages && ci[e].do_meshes == ci[0].do_meshes &&
              (c.init_cull_meshes != 0) == (contexts[0].init_cull_meshes != 0);
  }
  if (!fusable) {
    for (uint32_t e = 0; e < count; e++) {
      oxc_status st = oxc_cull_geometry(ctx, &frames[e], &contexts[e], hip_stream);
      if (st != OXC_OK) return st;
    }
    return OXC_OK;
  }
  // Several VIEWS of one scene (every element runs cull_meshes + cull_meshlets over the same meshes / transforms with its own camera --
  // shadow cascades, BASELINE configs[4]): the meshlet stage then runs once over the scene for all views (k_mv_test) instead of
  // once per view over that view's list.  Same outputs, element by element.
  bool multiview = ci[0].do_meshes && ci[0].do_meshlets && ci[0].M > 0;
  bool same_pos = true;
  for (uint32_t e = 1; multiview && e < count; e++) {
    multiview = ci[e].M == ci[0].M && frames[e].meshes_buffer.dptr == frames[0].meshes_buffer.dptr &&
                frames[e].transforms_world_buffer.dptr == frames[0].transforms_world_buffer.dptr && contexts[e].cull_flags == contexts[0].cull_flags;
    same_pos = same_pos && std::memcmp(contexts[e].cull_camera.position, contexts[0].cull_camera.position, 12) == 0;
  }
  // implicit_meshlet_instances: all elements or none, and only where nothing reads the records -- the multi-view meshlet stage
  uint32_t n_implicit = 0;
  for (uint32_t e = 0; e < count; e++) n_implicit += contexts[e].implicit_meshlet_instances ? 1u : 0u;
  if (n_implicit && (n_implicit != count || !multiview || ci[0].do_tris))
    return fail(ctx, OXC_INVALID_ARG, "cull_geometry_batch: implicit_meshlet_instances must be set on every element of a multi-view batch (same scene, cull_meshes in every element, no triangle stage)");
  const bool implicit_lists = n_implicit != 0;
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  ctx->shared.valid = false;  // (share_pass_tests: a batch may rebuild any list)
  for (uint32_t e = 0; e < count; e++) {
    oxc_status st = ensure_capacity(ctx, ci[e].M, ci[e].N, 0, e, static_cast<hipStream_t>(hip_stream));
    if (st != OXC_OK) return st;
  }
  OXC_ORDER(ctx, hip_stream);
  OXC_JOIN(ctx, hip_stream);  // the batched call uses lane 0's scratch in order on hip_stream
  if (!ctx->batch_dev) {
    if (stream_is_capturing(static_cast<hipStream_t>(hip_stream)))
      return fail(ctx, OXC_INVALID_ARG, "cull_geometry_batch: the first batched call allocates its argument block; make one un-captured call first");
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->batch_dev), sizeof(BatchElem) * kMaxBatch);
    if (e != hipSuccess) return fail(ctx, OXC_OUT_OF_MEMORY, "hipMalloc(batch argument block)", e);
    OXC_HIP(ctx, hipMemset(ctx->batch_dev, 0, sizeof(BatchElem) * kMaxBatch));  // fields no plain-pipeline block uses stay 0
    // The fill runs on the NULL stream; `s` is usually a non-blocking stream, i.e. NOT ordered behind it.  Round 4 found the race the hard way:
    // in a fresh process the first hipMemset loads its kernel lazily, k_prepare_batch (on `s`) wrote the argument blocks first, the late fill
    // zeroed them and the next kernel dereferenced null -- "memory access fault on address (nil)" in 2 of ~25 bench runs, only on slow boxes.
    OXC_HIP(ctx, hipDeviceSynchronize());
  }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const uint32_t max_grid = ctx->num_cus * 8;
  const bool do_meshes = ci[0].do_meshes, do_meshlets = ci[0].do_meshlets, do_tris = ci[0].do_tris;
  bool mv_expand_async = false;
  // The MeshletInstance expansion of a batched call: in order on `s`, or (multi-view, nothing of the call reads the records) on the context's
  // lowest-priority side stream, forked from `s` where this is called and joined at the end of the call.
  auto launch_mv_expand = [&](uint32_t count_, uint32_t g_expand_, uint32_t cap_, bool on_side) -> oxc_status {
    hipStream_t es = s;
    if (on_side) {
      if (!ctx->mv_side) {  // its own stream, at the lowest priority: the store stream yields wave slots to the meshlet stage's launches
        int lo = 0, hi = 0;
        OXC_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        OXC_HIP(ctx, hipStreamCreateWithPriority(&ctx->mv_side, hipStreamNonBlocking, lo));
      }
      if (!ctx->mv_fork) OXC_HIP(ctx, hipEventCreateWithFlags(&ctx->mv_fork, hipEventDisableTiming));
      if (!ctx->mv_join) OXC_HIP(ctx, hipEventCreateWithFlags(&ctx->mv_join, hipEventDisableTiming));
      OXC_HIP(ctx, hipEventRecord(ctx->mv_fork, s));
      OXC_HIP(ctx, hipStreamWaitEvent(ctx->mv_side, ctx->mv_fork, 0));
      es = ctx->mv_side;
    }
    {
      KernelTimer t(ctx, OXC_K_MESHES_EXPAND, es);
      // (beside the meshlet stage the expansion takes mv_expand_async blocks per CU, not every wave slot: the small set-up launches queue behind it otherwise)
      const uint32_t ecap = on_side ? std::max(1u, ctx->num_cus * ctx->mv_expand_async / count_) : cap_;
      launch_expand_batch(ctx->batch_dev, count_, std::min(g_expand_, ecap), es);
    }
    if (on_side) OXC_HIP(ctx, hipEventRecord(ctx->mv_join, es));
    return OXC_OK;
  };
  BatchCore cores[kMaxBatch];
  std::memset(cores, 0, sizeof cores);
  uint32_t g_prep = 1, g_expand = 1, g_test = 1, g_emit = 1, g_ttest = 1, g_temit = 1;
  for (uint32_t e = 0; e < count; e++) {
    const oxc_prepared_frame* f = &frames[e];
    oxc_cull_geometry_context* c = &contexts[e];
    oxc_ctx::Lane& L = ctx->lane[e];
    const uint32_t M = ci[e].M, N = ci[e].N;
    uint32_t* slot = next_slot(ctx);
    uint32_t *vis, *meshlets_cmd;
    uint32_t n_host = 0;
    if (c->init_cull_meshes) {
      vis = slot + SLOT_VIS;
      meshlets_cmd = slot + SLOT_MESHLETS_CMD;
      c->visibility_buffer = {vis, 12};
      c->cull_meshlets_cmd_buffer = {meshlets_cmd, 12};
    } else {
      vis = static_cast<uint32_t*>(c->visibility_buffer.dptr);
      meshlets_cmd = static_cast<uint32_t*>(c->cull_meshlets_cmd_buffer.dptr);
      if (vis >= ctx->slots && vis < ctx->slots + (size_t)kSlots * SLOT_U32S && ((vis - ctx->slots) % SLOT_U32S) == SLOT_VIS)
        n_host = ctx->seeded_total[(vis - ctx->slots) / SLOT_U32S];
      if (n_host > N) n_host = 0;
    }
    c->cull_triangles_cmd_buffer = {slot + SLOT_TRI_CMD, 12};
    c->draw_geometry_cmd_buffer = {slot + SLOT_DRAW_CMD, 20};
    const uint32_t m_chunks = cdiv(std::max(N, 1u), kMeshletChunk), t_chunks = cdiv(std::max(N, 1u), kTriChunk);

    // every field once; k_prepare_batch rebuilds the seven stage blocks from it on the device (expand_batch_core)
    BatchCore& k = cores[e];
    k.meshes = static_cast<const GpuMesh*>(f->meshes_buffer.dptr);
    k.transforms = static_cast<const float*>(f->transforms_world_buffer.dptr);
    k.mesh_instances = static_cast<GpuMeshInstance*>(f->mesh_instances_buffer.dptr);
    k.meshlet_instances = static_cast<GpuMeshletInstance*>(f->meshlet_instances_buffer.dptr);
    k.visible_out = static_cast<uint32_t*>(f->visible_meshlet_instances_indices_buffer.dptr);
    k.reordered_out = static_cast<uint32_t*>(f->reordered_indices_buffer.dptr);
    k.cache = L.cache;
    k.view_cache = L.view_cache;
    k.mesh_counts = L.mesh_counts;
    k.mesh_offsets = L.mesh_offsets;
    k.bits = L.bits;
    k.m_chunk_counts = L.m_chunk_counts;
    k.m_supers = L.m_supers;
    k.tri_masks = L.tri_masks;
    k.t_chunk_counts = L.t_chunk_counts;
    k.t_supers = L.t_supers;
    k.slot = slot;
    k.vis = vis;
    k.meshlets_cmd = meshlets_cmd;
    k.n_supers_meshlets = cdiv(cdiv(std::max(N, 1u), 64u), kChunksPerSuper);
    k.n_supers_tris = cdiv(t_chunks, kChunksPerSuper);
    k.mesh_instance_count = M;
    k.cull_flags = c->cull_flags;
    k.do_cull_meshes = do_meshes ? 1u : 0u;
    k.init_vis = c->init_cull_meshes ? 1u : 0u;
    k.n_host = n_host;
    k.n_cap = N;
    k.count_meshlets = 64u * kPlainGroups;
    k.cam = c->cull_camera;
    g_prep = std::max(g_prep, std::min(cdiv(std::max(std::max(M * 8u, k.n_supers_tris), 1u), 256), max_grid));
    g_expand = std::max(g_expand, std::max(std::min(cdiv(M, 4), max_grid), 1u));
    g_test = std::max(g_test, std::min(m_chunks, max_grid));
    g_emit = std::max(g_emit, std::min(cdiv(std::max(N, 1u), kMeshletSpan), max_grid));
    g_ttest = std::max(g_ttest, std::min(t_chunks, max_grid));
    g_temit = std::max(g_temit, std::min(cdiv(std::max(N, 1u), kTriSpan), max_grid));
  }
  // grid.y = count; grid.x cap per element for the small stages (the meshlet test kernel takes its full grid, below)
  const uint32_t cap = std::max(max_grid / count, ctx->num_cus * 2);
  {
    // One launch hands over all element cores: a 4.5 KB kernarg segment.  A runtime that refuses a segment of that
    // size fails the launch synchronously; nothing has been enqueued then and the elements are culled one by one.
    static_assert(kBatchPerPrepare == kMaxBatch, "one prepare launch per batched call");
    BatchBlob blob;
    std::memset(&blob, 0, sizeof blob);
    blob.count = count;
    blob.first = 0;
    std::memcpy(blob.core, cores, sizeof(BatchCore) * count);
    hipError_t launch_err;
    {
      KernelTimer t(ctx, OXC_K_PREPARE, s);
      launch_prepare_batch(blob, ctx->batch_dev, g_prep, s);
      launch_err = hipGetLastError();
    }
    if (launch_err != hipSuccess) {
      for (uint32_t e = 0; e < count; e++) {
        oxc_status st = oxc_cull_geometry(ctx, &frames[e], &contexts[e], hip_stream);
        if (st != OXC_OK) return st;
      }
      return OXC_OK;
    }
  }
  if (do_meshes) {
    {
      KernelTimer t(ctx, OXC_K_MESHES_SCAN, s);
      launch_scan_batch(ctx->batch_dev, count, s);
    }
    if (!implicit_lists) {  // (implicit: the records have no reader in this call; the runs are written by k_mv_group)
      // Round 5: in the multi-view form nothing of this call reads the MeshletInstance records (k_mv_test walks the instances' bounds), so their
      // expansion -- a pure store stream, 466 MB per 16 views of a 10 M-meshlet scene -- runs on the context's side stream beside the
      // latency-bound set-up launches and the VALU-bound test of the meshlet stage, and is joined at the end of the call (fork / join by
      // events, capturable like async_triangles).  A triangle stage in the call reads the records: in order then.
      mv_expand_async = multiview && do_meshlets && !do_tris && ctx->mv_expand_async != 0;
      // (Round 6 measured the fork behind the meshlet stage's three set-up launches instead -- they are chains of dependent loads and take 78 us next to
      //  the store stream against 33 us alone -- : 0.2600 against 0.2615 ms per 16 views on one box, 0.2616 against 0.2568 on another; the test and the
      //  emit then share the machine with the whole store stream.  The fork stays here.)
      oxc_status est = launch_mv_expand(count, g_expand, cap, mv_expand_async);
      if (est != OXC_OK) return est;
    }
  }
  if (do_meshlets && multiview) {
    const uint32_t Mv = ci[0].M;
    // scratch: view table | groups [M][views] | chunks per instance, their prefix [M] | totals | step list | per view: vchunks, vchunk0 [M],
    // ballots [chunks][4], counts, idbase [chunks], supers, totals
    uint64_t off = 0;
    auto carve = [&](uint64_t bytes) {
      uint64_t o = off;
      off = align_up(off + bytes, 256);
      return o;
    };
    uint32_t vchunks_max[kMaxBatch];
    uint64_t steps_max = 0;
    uint32_t max_vchunks = 0;
    for (uint32_t e = 0; e < count; e++) {
      vchunks_max[e] = cdiv(std::max(ci[e].N, 1u), 256u) + Mv;  // sum over instances of ceil(count / 256) <= N / 256 + M
      steps_max += vchunks_max[e];
      max_vchunks = std::max(max_vchunks, vchunks_max[e]);
    }
    const uint64_t o_table = carve(sizeof(MvView) * kMaxBatch);
    const uint64_t o_groups = carve((uint64_t)Mv * count * sizeof(MvGroup));
    const uint64_t o_gch = carve((uint64_t)Mv * 4), o_st0 = carve((uint64_t)Mv * 4), o_tot = carve(64);
    const uint64_t o_steps = carve(steps_max * 8);
    uint64_t o_vch[kMaxBatch], o_vc0[kMaxBatch], o_bits[kMaxBatch], o_cnt[kMaxBatch], o_idb[kMaxBatch], o_sup[kMaxBatch], o_vtot[kMaxBatch];
    for (uint32_t e = 0; e < count; e++) {
      o_vch[e] = carve((uint64_t)Mv * 4);
      o_vc0[e] = carve((uint64_t)Mv * 4);
      o_bits[e] = carve((uint64_t)vchunks_max[e] * 32);
      o_cnt[e] = carve((uint64_t)vchunks_max[e] * 4);
      o_idb[e] = carve((uint64_t)vchunks_max[e] * 4);
      o_sup[e] = carve((uint64_t)cdiv(vchunks_max[e], kChunksPerSuper) * 4 * kSuperStride);
      o_vtot[e] = carve(64);
    }
    if (off > ctx->mv_arena_bytes) {
      if (stream_is_capturing(s)) return fail(ctx, OXC_INVALID_ARG, "cull_geometry_batch: the multi-view scratch must grow but the stream is being captured; make one un-captured call of this size first");
      OXC_HIP(ctx, hipDeviceSynchronize());
      if (ctx->mv_arena) OXC_HIP(ctx, hipFree(ctx->mv_arena));
      ctx->mv_arena = nullptr;
      ctx->mv_arena_bytes = 0;
      hipError_t me = hipMalloc(&ctx->mv_arena, off);
      if (me != hipSuccess) return fail(ctx, OXC_OUT_OF_MEMORY, "hipMalloc(multi-view scratch)", me);
      ctx->mv_arena_bytes = off;
    }
    char* const mb = static_cast<char*>(ctx->mv_arena);
    MvArgs ma;
    std::memset(&ma, 0, sizeof ma);
    ma.views = count;
    ma.M = Mv;
    ma.same_pos = same_pos ? 1u : 0u;
    ma.dev = reinterpret_cast<MvView*>(mb + o_table);
    ma.groups = reinterpret_cast<MvGroup*>(mb + o_groups);
    ma.grp_chunks = reinterpret_cast<uint32_t*>(mb + o_gch);
    ma.inst_step0 = reinterpret_cast<uint32_t*>(mb + o_st0);
    ma.step_total = reinterpret_cast<uint32_t*>(mb + o_tot);
    ma.steps = reinterpret_cast<uint2*>(mb + o_steps);
    ma.tickets = ctx->lane[0].m_tickets;
    MvBlob blob;
    std::memset(&blob, 0, sizeof blob);
    for (uint32_t e = 0; e < count; e++) {
      MvView& w = blob.v[e];
      oxc_ctx::Lane& L = ctx->lane[e];
      w.rows = L.cache;
      w.mesh_counts = L.mesh_counts;
      w.mesh_offsets = L.mesh_offsets;
      w.vchunks = reinterpret_cast<uint32_t*>(mb + o_vch[e]);
      w.vchunk0 = reinterpret_cast<uint32_t*>(mb + o_vc0[e]);
      w.bits = reinterpret_cast<uint64_t*>(mb + o_bits[e]);
      w.counts = reinterpret_cast<uint32_t*>(mb + o_cnt[e]);
      w.idbase = reinterpret_cast<uint32_t*>(mb + o_idb[e]);
      w.supers = reinterpret_cast<uint32_t*>(mb + o_sup[e]);
      w.scan_total = reinterpret_cast<uint32_t*>(mb + o_vtot[e]);
      w.out = static_cast<uint32_t*>(frames[e].visible_meshlet_instances_indices_buffer.dptr);
      w.runs = static_cast<uint32_t*>(contexts[e].meshlet_instance_runs_buffer.dptr);
      w.tri_cmd = cores[e].slot + SLOT_TRI_CMD;
      w.n_cap = ci[e].N;
      w.n_supers = cdiv(vchunks_max[e], kChunksPerSuper);
      std::memcpy(w.cam_pos, contexts[e].cull_camera.position, 12);
    }
    {
      KernelTimer t(ctx, OXC_K_MULTIVIEW_SETUP, s);
      launch_mv_setup(ma, blob, std::max(1u, std::min(cdiv(Mv, 16u), max_grid)), s);  // 16 lanes per mesh instance
    }
    {
      KernelTimer t(ctx, OXC_K_MESHLETS_TEST, s);
      launch_mv_test(ma, ctx->num_cus, s);
    }
    {
      KernelTimer t(ctx, OXC_K_MESHLETS_EMIT, s);
      launch_mv_emit(ma, max_vchunks, max_grid, s);
    }
    if (mv_expand_async) OXC_HIP(ctx, hipStreamWaitEvent(s, ctx->mv_join, 0));  // join: the records are part of what the call leaves behind on `s`
  } else if (do_meshlets) {
    {
      KernelTimer t(ctx, OXC_K_MESHLETS_TEST, s);
      // At 64 VGPRs (8 waves/SIMD) the test kernel wants every block it can get: with 16 x 1M meshlets, blocks per element
      // 384 -> 74.6 us, 512 -> 73.7, 768 -> 73.2, 1024 -> 70.7, 2048 -> 70.0 (>= 977 blocks: one 1024-meshlet step per block)
      launch_meshlets_test_batch(ctx->batch_dev, count, g_test, s);
    }
    KernelTimer t(ctx, OXC_K_MESHLETS_EMIT, s);
    launch_meshlets_emit_batch(ctx->batch_dev, count, std::min(g_emit, cap), s);
  }
  if (do_tris) {
    {
      KernelTimer t(ctx, OXC_K_TRIANGLES_TEST, s);
      launch_tris_test_batch(ctx->batch_dev, count, std::min(g_ttest, cap), s);
    }
    KernelTimer t(ctx, OXC_K_TRIANGLES_EMIT, s);
    launch_tris_emit_batch(ctx->batch_dev, count, std::min(g_temit, cap), s);
  }
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_read_counters(oxc_ctx* ctx, const oxc_cull_geometry_context* c, oxc_counters* out, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!c || !out) return fail(ctx, OXC_INVALID_ARG, "read_counters: null argument");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  OXC_JOIN(ctx, hip_stream);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  uint32_t vis[3] = {0, 0, 0}, mc[3] = {0, 0, 0}, tc[3] = {0, 0, 0}, dc[5] = {0, 0, 0, 0, 0};
  if (c->visibility_buffer.dptr) OXC_HIP(ctx, hipMemcpyAsync(vis, c->visibility_buffer.dptr, 12, hipMemcpyDeviceToHost, s));
  if (c->cull_meshlets_cmd_buffer.dptr) OXC_HIP(ctx, hipMemcpyAsync(mc, c->cull_meshlets_cmd_buffer.dptr, 12, hipMemcpyDeviceToHost, s));
  if (c->cull_triangles_cmd_buffer.dptr) OXC_HIP(ctx, hipMemcpyAsync(tc, c->cull_triangles_cmd_buffer.dptr, 12, hipMemcpyDeviceToHost, s));
  if (c->draw_geometry_cmd_buffer.dptr) OXC_HIP(ctx, hipMemcpyAsync(dc, c->draw_geometry_cmd_buffer.dptr, 20, hipMemcpyDeviceToHost, s));
  OXC_HIP(ctx, hipStreamSynchronize(s));
  out->total_visible_meshlet_instances = vis[0];
  out->early_visible_meshlet_instances = vis[1];
  out->late_visible_meshlet_instances = vis[2];
  out->cull_meshlets_cmd_x = mc[0];
  out->cull_triangles_cmd_x = tc[0];
  out->draw_index_count = dc[0];
  // wide_triangle_index = 2 only: a call that emitted more indices than VkDrawIndexedIndirectCommand.indexCount can count zeroed instanceCount
  // (the command draws nothing); every other path leaves the 1 this library initialises it with
  if (c->draw_geometry_cmd_buffer.dptr && c->wide_triangle_index == 2u && (!c->stages || (c->stages & OXC_STAGE_TRIANGLES)) && dc[1] == 0u)
    return fail(ctx, OXC_INVALID_ARG, "read_counters: the call emitted more than 2^32 - 1 indices (pairs): draw_index_count wrapped, instanceCount was set to 0; cull fewer meshlets per call");
  return OXC_OK;
}

oxc_status oxc_profile_begin(oxc_ctx* ctx) {
  if (!ctx) return OXC_INVALID_ARG;
  ctx->recs.clear();
  ctx->profiling = true;
  return OXC_OK;
}

oxc_status oxc_profile_end(oxc_ctx* ctx, oxc_kernel_times* out) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!out) return fail(ctx, OXC_INVALID_ARG, "profile_end: null out");
  ctx->profiling = false;
  std::memset(out, 0, sizeof *out);
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_HIP(ctx, hipDeviceSynchronize());
  for (auto& r : ctx->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.id >= 0 && r.id < OXC_K_COUNT) {
      out->total_ms[r.id] += ms;
      out->launches[r.id] += 1;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->recs.clear();
  // cost of an empty event pair on an idle stream (subtract per launch)
  hipEvent_t a, b;
  OXC_HIP(ctx, hipEventCreate(&a));
  OXC_HIP(ctx, hipEventCreate(&b));
  double acc = 0;
  const int reps = 32;
  for (int i = 0; i < reps; i++) {
    OXC_HIP(ctx, hipEventRecord(a, nullptr));
    OXC_HIP(ctx, hipEventRecord(b, nullptr));
    OXC_HIP(ctx, hipEventSynchronize(b));
    float ms = 0.f;
    OXC_HIP(ctx, hipEventElapsedTime(&ms, a, b));
    acc += ms;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  out->empty_pair_ms = acc / reps;
  return OXC_OK;
}

oxc_status oxc_stream_read_probe(oxc_ctx* ctx, const void* dptr, uint64_t bytes, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!dptr || (reinterpret_cast<uintptr_t>(dptr) & 15u)) return fail(ctx, OXC_INVALID_ARG, "stream_read_probe: pointer must be 16-byte aligned");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  launch_stream_read(dptr, bytes, ctx->sink, ctx->num_cus * 8, static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_debug_decode_bounds(oxc_ctx* ctx, const void* bounds_dptr, uint32_t n, float* out10_dptr, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!bounds_dptr || !out10_dptr) return fail(ctx, OXC_INVALID_ARG, "debug_decode_bounds: null pointer");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  if (n) launch_debug_decode_bounds(bounds_dptr, n, out10_dptr, static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

constexpr uint32_t kBoundsChunk = 1u << 18;  // meshlets per gather/cone launch pair: bounds the normals scratch at 192 MiB

oxc_status oxc_build_meshlet_bounds(oxc_ctx* ctx, const oxc_meshlet_bounds_desc* d, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!d || d->struct_size != sizeof(oxc_meshlet_bounds_desc)) return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: bad desc / struct_size");
  if (!d->mesh_bounds.dptr || d->mesh_bounds.bytes < 24) return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: mesh_bounds must hold 6 floats");
  if (d->meshlet_count) {
    if (!d->positions.dptr || !d->meshlets.dptr || !d->indirect_vertex_indices.dptr || !d->local_triangle_indices.dptr || !d->meshlet_bounds.dptr)
      return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: null input / output buffer");
    if (d->meshlets.bytes < (uint64_t)d->meshlet_count * sizeof(GpuMeshlet) || d->meshlet_bounds.bytes < (uint64_t)d->meshlet_count * 16u)
      return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: meshlets / meshlet_bounds smaller than meshlet_count records");
  }
  if (d->positions.dptr && d->positions.bytes < (uint64_t)d->vertex_count * 12u) return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: positions smaller than vertex_count float3");
  if (d->quantized_positions.dptr && (!d->positions.dptr || d->quantized_positions.bytes < (uint64_t)d->vertex_count * 8u))
    return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: quantized_positions smaller than vertex_count u16x4");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  // scratch: [boxes: 24 B per meshlet][fold partials: 256 x 12 words][normals: 768 B per meshlet of one chunk][counts]
  const uint32_t want = std::max(d->meshlet_count, 1u);
  if (want > ctx->bounds_scratch_cap) {
    if (stream_is_capturing(static_cast<hipStream_t>(hip_stream)))
      return fail(ctx, OXC_INVALID_ARG, "build_meshlet_bounds: scratch must grow but the stream is being captured; make one un-captured call of this size first");
    OXC_HIP(ctx, hipDeviceSynchronize());  // in-flight work may still use the old scratch
    if (ctx->bounds_scratch) OXC_HIP(ctx, hipFree(ctx->bounds_scratch));
    ctx->bounds_scratch = nullptr;
    ctx->bounds_scratch_cap = 0;
    const size_t bytes = align_up((size_t)want * 24u, 256) + 256u * 48u + (size_t)std::min(kBoundsChunk, want) * (768u + 4u);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->bounds_scratch), bytes);
    if (e != hipSuccess) return fail(ctx, OXC_OUT_OF_MEMORY, "hipMalloc(bounds scratch)", e);
    ctx->bounds_scratch_cap = want;
  }
  const uint32_t chunk = std::min(kBoundsChunk, ctx->bounds_scratch_cap);
  float* fold = ctx->bounds_scratch + align_up((size_t)ctx->bounds_scratch_cap * 24u, 256) / 4;
  float* normals = fold + 256 * 12;
  {
    KernelTimer t(ctx, OXC_K_MESHLET_BOUNDS, static_cast<hipStream_t>(hip_stream));
    launch_build_meshlet_bounds(static_cast<const float*>(d->positions.dptr), d->vertex_count, d->meshlets.dptr, d->meshlet_count,
                                static_cast<const uint32_t*>(d->indirect_vertex_indices.dptr), static_cast<const uint8_t*>(d->local_triangle_indices.dptr),
                                d->meshlet_bounds.dptr, static_cast<float*>(d->mesh_bounds.dptr), d->quantized_positions.dptr, ctx->bounds_scratch, normals,
                                reinterpret_cast<uint32_t*>(normals + (size_t)chunk * 192), fold, chunk, ctx->num_cus * 8,
                                static_cast<hipStream_t>(hip_stream));
  }
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_quantize_vertex_streams(oxc_ctx* ctx, const oxc_vertex_streams_desc* d, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!d || d->struct_size != sizeof(oxc_vertex_streams_desc)) return fail(ctx, OXC_INVALID_ARG, "quantize_vertex_streams: bad desc / struct_size");
  const uint64_t n = d->vertex_count;
  struct Stream {
    const oxc_buffer *in, *out;
    uint32_t in_stride, out_stride;
    const char* what;
  } streams[3] = {{&d->positions, &d->quantized_positions, 12, 8, "quantize_vertex_streams: positions / quantized_positions too small or null"},
                  {&d->normals, &d->quantized_normals, 12, 4, "quantize_vertex_streams: normals / quantized_normals too small or null"},
                  {&d->texcoords, &d->quantized_texcoords, 8, 4, "quantize_vertex_streams: texcoords / quantized_texcoords too small or null"}};
  for (const Stream& st : streams)
    if (st.in->dptr && n && (!st.out->dptr || st.in->bytes < n * st.in_stride || st.out->bytes < n * st.out_stride)) return fail(ctx, OXC_INVALID_ARG, st.what);
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  launch_quantize_vertex_streams(static_cast<const float*>(d->positions.dptr), static_cast<const float*>(d->normals.dptr),
                                 static_cast<const float*>(d->texcoords.dptr), d->vertex_count, d->quantized_positions.dptr, d->quantized_normals.dptr,
                                 d->quantized_texcoords.dptr, ctx->num_cus * 8, static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

// ---- mesh blob (host arithmetic only) --------------------------------------------------------------
oxc_status oxc_mesh_blob_layout_of(const oxc_mesh_blob_desc* d, oxc_mesh_blob_layout* out) {
  if (!d || !out || d->struct_size != sizeof(oxc_mesh_blob_desc)) return OXC_INVALID_ARG;
  if (d->lod_count == 0 || d->lod_count > OXC_MESH_MAX_LODS) return OXC_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  uint64_t size = 0;
  auto append = [&size](uint64_t bytes, uint64_t alignment) {  // blob_append, AssetManager_GLTF.cpp:466-474
    const uint64_t offset = (size + alignment - 1) / alignment * alignment;
    size = offset + bytes;
    return offset;
  };
  const uint64_t V = d->vertex_count;
  out->vertex_positions = append(V * 8, 8);  // :592-596
  out->vertex_normals = append(V * 4, 4);
  if (d->has_texture_coords) out->texture_coords = append(V * 4, 4);
  for (uint32_t i = 0; i < d->lod_count; i++) {  // :748-752
    const oxc_mesh_lod_counts& c = d->lods[i];
    oxc_mesh_lod_offsets& o = out->lods[i];
    o.indices = append((uint64_t)c.indices_count * 4, 8);
    o.meshlets = append((uint64_t)c.meshlet_count * sizeof(GpuMeshlet), 8);
    o.meshlet_bounds = append((uint64_t)c.meshlet_count * 16, 8);
    o.local_triangle_indices = append(c.local_triangle_indices_count, 8);
    o.indirect_vertex_indices = append((uint64_t)c.indirect_vertex_indices_count * 4, 4);
  }
  out->lod_metadata_offset = (size + 7) / 8 * 8;  // :768-769
  out->size = out->lod_metadata_offset + (uint64_t)d->lod_count * sizeof(GpuMeshLOD);
  return OXC_OK;
}

oxc_status oxc_mesh_blob_finalize(const oxc_mesh_blob_desc* d, const oxc_mesh_blob_layout* l, uint64_t device_address, void* blob, uint64_t blob_bytes,
                                  const float mesh_bounds[6], void* out_gpu_mesh) {
  if (!d || !l || !blob || !mesh_bounds || !out_gpu_mesh || d->struct_size != sizeof(oxc_mesh_blob_desc)) return OXC_INVALID_ARG;
  if (d->lod_count == 0 || d->lod_count > OXC_MESH_MAX_LODS) return OXC_INVALID_ARG;
  if (blob_bytes < l->size || l->lod_metadata_offset + (uint64_t)d->lod_count * sizeof(GpuMeshLOD) > l->size) return OXC_INVALID_ARG;
  GpuMeshLOD table[OXC_MESH_MAX_LODS] = {};
  for (uint32_t i = 0; i < d->lod_count; i++) {  // AssetManager_GLTF.cpp:787-794 + the counts of :754-758
    const oxc_mesh_lod_counts& c = d->lods[i];
    const oxc_mesh_lod_offsets& o = l->lods[i];
    GpuMeshLOD& t = table[i];
    t.indices = device_address + o.indices;
    t.meshlets = device_address + o.meshlets;
    t.meshlet_bounds = device_address + o.meshlet_bounds;
    t.local_triangle_indices = device_address + o.local_triangle_indices;
    t.indirect_vertex_indices = device_address + o.indirect_vertex_indices;
    t.indices_count = c.indices_count;
    t.meshlet_count = c.meshlet_count;
    t.meshlet_bounds_count = c.meshlet_count;
    t.local_triangle_indices_count = c.local_triangle_indices_count;
    t.indirect_vertex_indices_count = c.indirect_vertex_indices_count;
    t.error = c.error;
  }
  std::memcpy(static_cast<uint8_t*>(blob) + l->lod_metadata_offset, table, (size_t)d->lod_count * sizeof(GpuMeshLOD));  // :796-800
  GpuMesh m = {};
  m.vertex_positions = device_address + l->vertex_positions;  // :780-785
  m.vertex_normals = device_address + l->vertex_normals;
  m.texture_coords = d->has_texture_coords ? device_address + l->texture_coords : 0;
  m.vertex_count = d->vertex_count;
  m.lod_count = d->lod_count;
  m.lods = device_address + l->lod_metadata_offset;
  for (int k = 0; k < 3; k++) m.aabb_center[k] = mesh_bounds[k], m.aabb_extent[k] = mesh_bounds[3 + k];
  std::memcpy(out_gpu_mesh, &m, sizeof(m));
  return OXC_OK;
}

oxc_status oxc_generate_hpb(oxc_ctx* ctx, oxc_buffer page_table, const oxc_image_array_u8* h, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!h || !h->dptr) return fail(ctx, OXC_INVALID_ARG, "generate_hpb: null hpb_attachment");
  if (h->levels == 0 || h->levels > 13 || h->width == 0 || h->height == 0) return fail(ctx, OXC_INVALID_ARG, "generate_hpb: bad extent / level count");
  const uint64_t pages = (uint64_t)h->width * h->height * h->layers;
  if (pages && (!page_table.dptr || page_table.bytes < pages * 4u)) return fail(ctx, OXC_INVALID_ARG, "generate_hpb: virtual_page_table smaller than width*height*layers u32");
  for (uint32_t k = 0; k < h->levels; k++)
    if (h->level_offset[k] > 0xFFFFFFFFull) return fail(ctx, OXC_INVALID_ARG, "generate_hpb: level offsets must fit 32 bits");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  launch_generate_hpb(static_cast<const uint32_t*>(page_table.dptr), static_cast<uint8_t*>(h->dptr), h->width, h->height, h->layers, h->levels, h->level_offset,
                      static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_cull_terrain(oxc_ctx* ctx, oxc_terrain_context* c, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!c || c->struct_size != sizeof(oxc_terrain_context)) return fail(ctx, OXC_INVALID_ARG, "cull_terrain: bad context / struct_size");
  const uint64_t total64 = (uint64_t)c->patch_count[0] * c->patch_count[1];
  if (c->patch_count[0] == 0 || c->patch_count[1] == 0 || total64 > (1u << 24)) return fail(ctx, OXC_INVALID_ARG, "cull_terrain: patch_count must be 1..2^24 patches");
  const uint32_t total = (uint32_t)total64;
  const oxc_image& mm = c->patch_minmax_attachment;
  if (!mm.dptr || mm.width != c->patch_count[0] || mm.height != c->patch_count[1]) return fail(ctx, OXC_INVALID_ARG, "cull_terrain: patch_minmax_attachment must be patch_count.x x patch_count.y RG32F");
  if (!c->visible_patches_buffer.dptr || c->visible_patches_buffer.bytes < (uint64_t)total * 4u) return fail(ctx, OXC_INVALID_ARG, "cull_terrain: visible_patches_buffer smaller than patch_total u32");
  if (!c->patch_visibility_mask_buffer.dptr || c->patch_visibility_mask_buffer.bytes < (uint64_t)cdiv(total, 32u) * 4u)
    return fail(ctx, OXC_INVALID_ARG, "cull_terrain: patch_visibility_mask_buffer smaller than ceil(patch_total / 32) words");
  const bool needs_hiz = (c->cull_flags & (OXC_CULL_TEST_OCCLUSION | OXC_CULL_LATE_PASS)) != 0u;
  const oxc_image& h = c->hiz_attachment;
  if (needs_hiz && (!h.dptr || h.levels == 0 || h.levels > 13 || h.width == 0 || h.height == 0)) return fail(ctx, OXC_INVALID_ARG, "cull_terrain: TestOcclusion / LatePass need hiz_attachment");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  if (total > 1024u) {  // the two-kernel form borrows the meshlet stage's ballot / count scratch (one ballot per 64, one count per 1024 patches)
    // emit_bits is indexed [block * 16 + wave]: 16 ballots per STARTED block of 1024 patches, so the request is rounded up to whole blocks
    // (lane.bits holds ceil(n / 64) ballots for a request of n)
    oxc_status cst = ensure_capacity(ctx, 0, cdiv(total, 1024u) * 1024u, 0, 0, static_cast<hipStream_t>(hip_stream));
    if (cst != OXC_OK) return cst;
  }
  OXC_ORDER(ctx, hip_stream);
  uint32_t* slot = next_slot(ctx);
  TerrainArgs a;
  std::memset(&a, 0, sizeof a);
  std::memcpy(a.pv, c->cull_camera.projection_view, 64);
  a.near_clip = c->cull_camera.near_clip;
  a.cull_flags = c->cull_flags;
  a.world_min[0] = c->world_min[0];
  a.world_min[1] = c->world_min[1];
  a.world_size[0] = c->world_size[0];
  a.world_size[1] = c->world_size[1];
  a.pcx = c->patch_count[0];
  a.pcy = c->patch_count[1];
  a.base_height = c->base_height;
  a.height_scale = c->height_scale;
  a.patch_minmax = reinterpret_cast<const float2*>(static_cast<const char*>(mm.dptr) + mm.level_offset[0]);
  if (needs_hiz) {
    a.hiz_data = static_cast<const float*>(h.dptr);
    a.hiz_w = h.width;
    a.hiz_h = h.height;
    a.hiz_levels = h.levels;
    for (uint32_t k = 0; k < h.levels; k++) a.hiz_level_off[k] = (uint32_t)(h.level_offset[k] / 4);
  }
  a.mask = static_cast<uint32_t*>(c->patch_visibility_mask_buffer.dptr);
  a.visible = static_cast<uint32_t*>(c->visible_patches_buffer.dptr);
  a.draw_cmd = slot + SLOT_DRAW_CMD;
  a.emit_bits = ctx->lane[0].bits;
  a.block_counts = ctx->lane[0].m_chunk_counts;
  c->cull_camera.mesh_instance_count = total;  // Terrain.cpp:171
  c->draw_cmd_buffer = {a.draw_cmd, 16};
  launch_cull_terrain(a, static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

// Big triangles (and clipped triangle ids) queued per draw; beyond that the producing lane rasterises the triangle itself and an
// overflow pass re-walks the index list for the crossing triangles the id queue could not hold (both slow, both correct).  240 MB of scratch, allocated by the first draw.
// oxc_debug_set_tuning(OXC_TUNE_RASTER_BIG_CAPACITY) before that first draw shrinks it so that the tests can reach the overflow paths with a small scene.
constexpr uint32_t kRasterBigCapacity = 1u << 22;

oxc_status oxc_draw_visbuffer(oxc_ctx* ctx, const oxc_prepared_frame* f, const oxc_draw_context* d, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!f || !d || d->struct_size != sizeof(oxc_draw_context)) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: bad frame / context / struct_size");
  if (d->width == 0 || d->height == 0 || d->width > 16384 || d->height > 16384) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: extent must be 1..16384");
  if (d->wide_triangle_index > 2u) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: wide_triangle_index must be 0, 1 or 2");
  const uint64_t n = (uint64_t)d->width * d->height;
  if (!d->visdepth_buffer.dptr || d->visdepth_buffer.bytes < n * 8u) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: visdepth_buffer smaller than width*height u64");
  if (!d->draw_geometry_cmd_buffer.dptr) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: draw_geometry_cmd_buffer is null (run oxc_cull_geometry with the triangle stage first)");
  if (!f->meshes_buffer.dptr || !f->transforms_world_buffer.dptr || !f->mesh_instances_buffer.dptr || !f->meshlet_instances_buffer.dptr ||
      !f->reordered_indices_buffer.dptr)
    return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: null PreparedFrame buffer");
  const oxc_image& dep = d->depth_attachment;
  if (dep.dptr && (dep.width != d->width || dep.height != d->height)) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: depth_attachment extent differs from the draw extent");
  if (d->visbuffer_attachment.dptr && d->visbuffer_attachment.bytes < n * 4u) return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: visbuffer_attachment smaller than width*height u32");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  OXC_JOIN(ctx, hip_stream);  // reads reordered_indices / the draw command
  if (!ctx->raster_scratch) {
    if (stream_is_capturing(static_cast<hipStream_t>(hip_stream)))
      return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: the first call allocates its scratch; make one un-captured call first");
    uint32_t cap = kRasterBigCapacity;
    if (ctx->raster_capacity_request) cap = std::min<uint32_t>(std::max<uint32_t>(ctx->raster_capacity_request, kBigSegs * 16), 1u << 24) / kBigSegs * kBigSegs;
    hipError_t e = hipMalloc(&ctx->raster_scratch, (size_t)cap * kTriSetupBytes + kRasterHeaderBytes + (size_t)cap * 4 + (size_t)cap * 2 * 8);
    if (e != hipSuccess) return fail(ctx, OXC_OUT_OF_MEMORY, "hipMalloc(raster scratch)", e);
    ctx->raster_capacity = cap;
  }
  if (f->mesh_instance_count > ctx->raster_rows_cap) {
    if (stream_is_capturing(static_cast<hipStream_t>(hip_stream)))
      return fail(ctx, OXC_INVALID_ARG, "draw_visbuffer: more mesh instances than any earlier call (scratch would grow); make one un-captured call first");
    OXC_HIP(ctx, hipDeviceSynchronize());  // in-flight draws may still read the old rows
    if (ctx->raster_rows) OXC_HIP(ctx, hipFree(ctx->raster_rows));
    ctx->raster_rows = nullptr;
    ctx->raster_rows_cap = 0;
    hipError_t e = hipMalloc(&ctx->raster_rows, (size_t)f->mesh_instance_count * sizeof(DrawRow));
    if (e != hipSuccess) return fail(ctx, OXC_OUT_OF_MEMORY, "hipMalloc(raster rows)", e);
    ctx->raster_rows_cap = f->mesh_instance_count;
  }
  DrawArgs a;
  std::memset(&a, 0, sizeof a);
  std::memcpy(a.pv, d->projection_view, 64);
  a.rows = static_cast<DrawRow*>(ctx->raster_rows);
  a.mesh_instance_count = f->mesh_instance_count;
  a.meshes = static_cast<const GpuMesh*>(f->meshes_buffer.dptr);
  a.transforms = static_cast<const float*>(f->transforms_world_buffer.dptr);
  a.mesh_instances = static_cast<const GpuMeshInstance*>(f->mesh_instances_buffer.dptr);
  a.meshlet_instances = static_cast<const GpuMeshletInstance*>(f->meshlet_instances_buffer.dptr);
  a.indices = static_cast<const uint32_t*>(f->reordered_indices_buffer.dptr);
  a.draw_cmd = static_cast<const uint32_t*>(d->draw_geometry_cmd_buffer.dptr);
  a.visdepth = static_cast<unsigned long long*>(d->visdepth_buffer.dptr);
  a.width = d->width;
  a.height = d->height;
  a.wide = d->wide_triangle_index;
  char* const rs = static_cast<char*>(ctx->raster_scratch);
  a.clip_count = reinterpret_cast<uint32_t*>(rs);
  a.tile_count = reinterpret_cast<uint32_t*>(rs) + 32;
  a.big_seg_counts = reinterpret_cast<uint32_t*>(rs + 256);
  const uint32_t cap = ctx->raster_capacity;
  a.big_seg_capacity = cap / kBigSegs;
  a.big_list = reinterpret_cast<TriSetup*>(rs + kRasterHeaderBytes);
  a.clip_capacity = cap;  // more clipped triangles than that in one draw: k_draw_clipped<RESCAN> finds them again
  a.clip_list = reinterpret_cast<uint32_t*>(rs + kRasterHeaderBytes + (size_t)cap * kTriSetupBytes);
  a.tile_capacity = cap * 2;
  a.tile_list = reinterpret_cast<uint2*>(rs + kRasterHeaderBytes + (size_t)cap * (kTriSetupBytes + 4));
  {
    KernelTimer t(ctx, OXC_K_DRAW_VISBUFFER, static_cast<hipStream_t>(hip_stream));
    launch_draw_visbuffer(a, d->clear != 0, dep.dptr ? reinterpret_cast<float*>(static_cast<char*>(dep.dptr) + dep.level_offset[0]) : nullptr,
                          static_cast<uint32_t*>(d->visbuffer_attachment.dptr), ctx->num_cus * 8, static_cast<hipStream_t>(hip_stream));
  }
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

// ---- RCCL through dlopen: the exchange entry points are the only users ----
struct OxcRcclId {  // ncclUniqueId: passed BY VALUE to ncclCommInitRank
  char internal[OXC_COMM_UNIQUE_ID_BYTES];
};
namespace {
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, OxcRcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) {
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
      r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    }
  }
  return (r.lib && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Broadcast) ? &r : nullptr;
}
oxc_status rccl_fail(oxc_ctx* ctx, const char* what, int code) {
  static thread_local std::string msg;
  Rccl* r = rccl();
  msg = std::string(what) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(code) : "RCCL error");
  ctx->last_error = msg;
  return OXC_RCCL_ERROR;
}
constexpr int kNcclUint8 = 1, kNcclUint32 = 3;  // ncclDataType_t values (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)
}  // namespace

oxc_status oxc_comm_unique_id(oxc_ctx* ctx, void* id128_host_out) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!id128_host_out) return fail(ctx, OXC_INVALID_ARG, "comm_unique_id: null output");
  Rccl* r = rccl();
  if (!r) return fail(ctx, OXC_RCCL_ERROR, "librccl.so could not be loaded (needed only for the multi-GPU exchange)");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  int rc = r->GetUniqueId(id128_host_out);
  return rc == 0 ? OXC_OK : rccl_fail(ctx, "ncclGetUniqueId", rc);
}

oxc_status oxc_comm_init(oxc_ctx* ctx, const void* id128_host, uint32_t rank, uint32_t world) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!id128_host || world == 0 || rank >= world) return fail(ctx, OXC_INVALID_ARG, "comm_init: null id or rank >= world");
  if (ctx->comm) return fail(ctx, OXC_INVALID_ARG, "comm_init: the context already has a communicator");
  Rccl* r = rccl();
  if (!r) return fail(ctx, OXC_RCCL_ERROR, "librccl.so could not be loaded (needed only for the multi-GPU exchange)");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OxcRcclId id;
  std::memcpy(&id, id128_host, sizeof id);
  int rc = r->CommInitRank(&ctx->comm, (int)world, id, (int)rank);
  if (rc != 0) {
    ctx->comm = nullptr;
    return rccl_fail(ctx, "ncclCommInitRank", rc);
  }
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return OXC_OK;
}

oxc_status oxc_comm_destroy(oxc_ctx* ctx) {
  if (!ctx) return OXC_INVALID_ARG;
  if (ctx->comm) {
    Rccl* r = rccl();
    if (r) (void)r->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
  }
  return OXC_OK;
}

oxc_status oxc_pack_counters(oxc_ctx* ctx, const oxc_cull_geometry_context* c, void* counts4_dptr, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!c || c->struct_size != sizeof(oxc_cull_geometry_context) || !counts4_dptr) return fail(ctx, OXC_INVALID_ARG, "pack_counters: bad context struct or null output");
  if (!c->visibility_buffer.dptr || !c->cull_triangles_cmd_buffer.dptr || !c->draw_geometry_cmd_buffer.dptr)
    return fail(ctx, OXC_INVALID_ARG, "pack_counters: the context has no counter buffers yet (no cull_geometry call has filled them in)");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  OXC_JOIN(ctx, hip_stream);
  launch_pack_counters(static_cast<const uint32_t*>(c->visibility_buffer.dptr), static_cast<const uint32_t*>(c->cull_triangles_cmd_buffer.dptr),
                       static_cast<const uint32_t*>(c->draw_geometry_cmd_buffer.dptr), static_cast<uint32_t*>(counts4_dptr), static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_pack_counters_batch(oxc_ctx* ctx, uint32_t count, const oxc_cull_geometry_context* cs, void* counts4_dptr, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!cs || !counts4_dptr || count == 0 || count > kMaxBatch) return fail(ctx, OXC_INVALID_ARG, "pack_counters_batch: 1..16 contexts and an output buffer");
  PackBlob b;
  std::memset(&b, 0, sizeof b);
  b.count = count;
  for (uint32_t e = 0; e < count; e++) {
    if (cs[e].struct_size != sizeof(oxc_cull_geometry_context)) return fail(ctx, OXC_INVALID_ARG, "pack_counters_batch: bad context struct");
    if (!cs[e].visibility_buffer.dptr || !cs[e].cull_triangles_cmd_buffer.dptr || !cs[e].draw_geometry_cmd_buffer.dptr)
      return fail(ctx, OXC_INVALID_ARG, "pack_counters_batch: a context has no counter buffers yet (no cull_geometry call has filled them in)");
    b.vis[e] = static_cast<const uint32_t*>(cs[e].visibility_buffer.dptr);
    b.tri_cmd[e] = static_cast<const uint32_t*>(cs[e].cull_triangles_cmd_buffer.dptr);
    b.draw_cmd[e] = static_cast<const uint32_t*>(cs[e].draw_geometry_cmd_buffer.dptr);
  }
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  OXC_JOIN(ctx, hip_stream);
  launch_pack_counters_batch(b, static_cast<uint32_t*>(counts4_dptr), static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

oxc_status oxc_exchange_counts(oxc_ctx* ctx, const void* counts4_dptr, void* all_counts_dptr, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, OXC_INVALID_ARG, "exchange_counts: oxc_comm_init has not been called");
  if (!counts4_dptr || !all_counts_dptr) return fail(ctx, OXC_INVALID_ARG, "exchange_counts: null buffer");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  int rc = rccl()->AllGather(counts4_dptr, all_counts_dptr, 4, kNcclUint32, ctx->comm, static_cast<hipStream_t>(hip_stream));
  return rc == 0 ? OXC_OK : rccl_fail(ctx, "ncclAllGather", rc);
}

oxc_status oxc_broadcast_hiz(oxc_ctx* ctx, const oxc_image* hiz, uint64_t total_bytes, uint32_t root, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, OXC_INVALID_ARG, "broadcast_hiz: oxc_comm_init has not been called");
  if (!hiz || !hiz->dptr || total_bytes == 0 || root >= ctx->comm_world) return fail(ctx, OXC_INVALID_ARG, "broadcast_hiz: null image, zero size or bad root");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  int rc = rccl()->Broadcast(hiz->dptr, hiz->dptr, (size_t)total_bytes, kNcclUint8, (int)root, ctx->comm, static_cast<hipStream_t>(hip_stream));
  return rc == 0 ? OXC_OK : rccl_fail(ctx, "ncclBroadcast", rc);
}

oxc_status oxc_broadcast_hiz_levels(oxc_ctx* ctx, const oxc_image* hiz, uint32_t first_level, uint64_t total_bytes, uint32_t root, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!ctx->comm) return fail(ctx, OXC_INVALID_ARG, "broadcast_hiz_levels: oxc_comm_init has not been called");
  if (!hiz || !hiz->dptr || first_level >= hiz->levels || hiz->levels > 13 || root >= ctx->comm_world) return fail(ctx, OXC_INVALID_ARG, "broadcast_hiz_levels: null image, bad level or bad root");
  const uint64_t begin = hiz->level_offset[first_level];
  if (total_bytes <= begin) return fail(ctx, OXC_INVALID_ARG, "broadcast_hiz_levels: total_bytes does not reach first_level");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  char* p = static_cast<char*>(hiz->dptr) + begin;
  int rc = rccl()->Broadcast(p, p, (size_t)(total_bytes - begin), kNcclUint8, (int)root, ctx->comm, static_cast<hipStream_t>(hip_stream));
  return rc == 0 ? OXC_OK : rccl_fail(ctx, "ncclBroadcast", rc);
}

oxc_status oxc_debug_project_aabb(oxc_ctx* ctx, const float* mvp16_host, float near_clip, const void* boxes6_dptr, uint32_t n, float* out7_dptr, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!mvp16_host || !boxes6_dptr || !out7_dptr) return fail(ctx, OXC_INVALID_ARG, "debug_project_aabb: null pointer");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  DebugProjectArgs a;
  std::memcpy(a.mvp, mvp16_host, 64);
  a.near_clip = near_clip;
  a.n = n;
  a.boxes6 = static_cast<const float*>(boxes6_dptr);
  a.out7 = out7_dptr;
  launch_debug_project_aabb(a, static_cast<hipStream_t>(hip_stream));
  OXC_HIP(ctx, hipGetLastError());
  return OXC_OK;
}

uint32_t oxc_debug_shared_tests_mode(const oxc_ctx* ctx) { return ctx ? ctx->last_share_mode : 0u; }
uint32_t oxc_debug_tri_loads_mode(const oxc_ctx* ctx) { return ctx ? ctx->last_tri_loads : 0u; }

oxc_status oxc_debug_set_tuning(oxc_ctx* ctx, uint32_t knob, uint32_t value) {
  if (!ctx) return OXC_INVALID_ARG;
  switch (knob) {
    case OXC_TUNE_ASYNC_MTEST_BLOCKS_PER_CU: ctx->async_mtest_per_cu = value; return OXC_OK;
    case OXC_TUNE_ASYNC_TRI_BLOCKS_PER_CU: ctx->async_tri_per_cu = value; return OXC_OK;
    case OXC_TUNE_TRI_BLOCKS_PER_CU:
      if (value == 0) return fail(ctx, OXC_INVALID_ARG, "set_tuning: the triangle grid needs at least one block per CU");
      ctx->tri_blocks_per_cu = value;
      return OXC_OK;
    case OXC_TUNE_MV_EXPAND_ASYNC: ctx->mv_expand_async = value; return OXC_OK;
    case OXC_TUNE_TRI_LOADS:
      if (value > 2u) return fail(ctx, OXC_INVALID_ARG, "set_tuning: OXC_TUNE_TRI_LOADS is 0 (by the scene), 1 (nt) or 2 (plain)");
      ctx->tri_loads = value;
      return OXC_OK;
    case OXC_TUNE_RASTER_BIG_CAPACITY:
      if (ctx->raster_scratch) return fail(ctx, OXC_INVALID_ARG, "set_tuning: the raster scratch is allocated by the first oxc_draw_visbuffer; set its capacity before");
      ctx->raster_capacity_request = value;
      return OXC_OK;
    default: return fail(ctx, OXC_INVALID_ARG, "set_tuning: unknown knob");
  }
}

oxc_status oxc_debug_count_occlusion_candidates(oxc_ctx* ctx, void* counters_dptr) {
  if (!ctx) return OXC_INVALID_ARG;
  ctx->dbg_occlusion = static_cast<uint32_t*>(counters_dptr);
  return OXC_OK;
}

oxc_status oxc_debug_read_u32(oxc_ctx* ctx, const void* dptr, uint32_t n, uint32_t* host_out, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!dptr || !host_out) return fail(ctx, OXC_INVALID_ARG, "debug_read_u32: null pointer");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  OXC_HIP(ctx, hipMemcpyAsync(host_out, dptr, (size_t)n * 4u, hipMemcpyDeviceToHost, s));
  OXC_HIP(ctx, hipStreamSynchronize(s));
  return OXC_OK;
}

oxc_status oxc_debug_raster_stats(oxc_ctx* ctx, uint32_t* host_out4, void* hip_stream) {
  if (!ctx) return OXC_INVALID_ARG;
  if (!host_out4) return fail(ctx, OXC_INVALID_ARG, "debug_raster_stats: null pointer");
  if (!ctx->raster_scratch) return fail(ctx, OXC_INVALID_ARG, "debug_raster_stats: no oxc_draw_visbuffer call on this context yet");
  OXC_HIP(ctx, hipSetDevice(ctx->device));
  OXC_ORDER(ctx, hip_stream);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  std::vector<uint32_t> h(kRasterHeaderBytes / 4);
  OXC_HIP(ctx, hipMemcpyAsync(h.data(), ctx->raster_scratch, kRasterHeaderBytes, hipMemcpyDeviceToHost, s));
  OXC_HIP(ctx, hipStreamSynchronize(s));
  const uint32_t seg_cap = ctx->raster_capacity / kBigSegs;
  uint64_t big = 0;
  uint32_t overflowed = 0;
  for (uint32_t k = 0; k < kBigSegs; k++) {
    const uint32_t c = h[64 + k * kBigSegStride];
    big += c;
    overflowed += c > seg_cap ? 1u : 0u;
  }
  host_out4[0] = (uint32_t)std::min<uint64_t>(big, 0xFFFFFFFFull);
  host_out4[1] = h[0];
  host_out4[2] = h[32];
  host_out4[3] = overflowed;
  return OXC_OK;
}

}  // extern "C"
