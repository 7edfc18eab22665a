/* oxcull_debug.h -- test, harness and measurement hooks of liboxcull.so.
 *
 * NOT part of the drop-in boundary: include/oxcull.h is what an engine binds (INTEGRATION.md sections 1-2, oxylus_amd/host/RendererInstance.hpp
 * includes only that).  The entry points below exist for this repo's tests (known-answer sweeps of the device's decode and projection
 * routines, read-backs, the overflow paths of the rasteriser), for bench.py (per-kernel HIP-event timing, the streaming-read ceiling,
 * SURVEY 8d's occlusion-candidate count) and for tools/kbench.py (grid caps that the A/B measurements move).  They are exported by the
 * same library; tests/test_abi.py checks both headers against the export table. */
#ifndef OXCULL_DEBUG_H
#define OXCULL_DEBUG_H
#include "oxcull.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Streaming-read probe: sums `bytes` of device memory with 16 B/lane loads (the measured HBM
 * ceiling SURVEY 8d asks to report next to the 8 TB/s spec figure). */
oxc_status oxc_stream_read_probe(oxc_ctx* ctx, const void* dptr, uint64_t bytes, void* hip_stream);

/* ---- per-kernel timing (bench / profiling; off by default) ----
 * Between oxc_profile_begin and oxc_profile_end every kernel the context launches is bracketed
 * by a pair of hipEvents on the caller's stream.  oxc_profile_end synchronises and returns, per
 * kernel, the launch count and the summed event-to-event time, plus the measured cost of an
 * empty event pair (`empty_pair_ms`, to be subtracted per launch). */
enum {
  OXC_K_PREPARE = 0,
  OXC_K_MESHES_SCAN = 1,
  OXC_K_MESHES_EXPAND = 2,
  OXC_K_MESHLETS_TEST = 3,  /* plain / early-pass variants */
  OXC_K_MESHLETS_EMIT = 4,
  OXC_K_TRIANGLES_TEST = 5,
  OXC_K_TRIANGLES_EMIT = 6,
  OXC_K_HIZ = 7,
  OXC_K_MESHLETS_TEST_LATE = 8, /* the LatePass instantiations, timed apart: their candidate sets differ */
  OXC_K_MESHLETS_EMIT_LATE = 9,
  OXC_K_TRIANGLES_TEST_LATE = 10,
  OXC_K_TRIANGLES_EMIT_LATE = 11,
  OXC_K_DRAW_VISBUFFER = 12, /* every launch of one oxc_draw_visbuffer call (clear, setup, clipped, big, resolve) */
  OXC_K_MESHLET_BOUNDS = 13, /* oxc_build_meshlet_bounds */
  OXC_K_MULTIVIEW_SETUP = 14, /* batched views of one scene: view groups per mesh instance, chunk numbering, step list (three small launches) */
  OXC_K_COUNT = 16
};
typedef struct oxc_kernel_times {
  double total_ms[16];
  uint32_t launches[16];
  double empty_pair_ms;
} oxc_kernel_times;
oxc_status oxc_profile_begin(oxc_ctx* ctx);
oxc_status oxc_profile_end(oxc_ctx* ctx, oxc_kernel_times* out);

/* Test hook: decode n GPU::MeshletBounds records with the device's dequantize_half / s8/127
 * routines into 10 floats each {center.xyz, extent.xyz, cone_axis.xyz, cone_cutoff}
 * (scene.slang:401-435) -- lets the known-answer tests sweep all 65536 halfs and 256 s8s. */
oxc_status oxc_debug_decode_bounds(oxc_ctx* ctx, const void* bounds_dptr, uint32_t n, float* out10_dptr,
                                   void* hip_stream);

/* Test hook: project_aabb (cull.slang:12-47) of n boxes {center.xyz, extent.xyz} with one matrix: out7 = {min.u, min.v,
 * min.z, max.u, max.v, max.z, returned ? 1 : 0} per box -- lets the tests compare the device's division fast path with IEEE
 * division bit for bit. */
oxc_status oxc_debug_project_aabb(oxc_ctx* ctx, const float* mvp16_host, float near_clip, const void* boxes6_dptr, uint32_t n, float* out7_dptr,
                                  void* hip_stream);

/* Harness helper: copy n u32 from device memory (e.g. a callee-owned indirect command) to the host; synchronises the stream. */
oxc_status oxc_debug_read_u32(oxc_ctx* ctx, const void* dptr, uint32_t n, uint32_t* host_out, void* hip_stream);
/* Test hook: what share_pass_tests did in the context's last oxc_cull_geometry call -- 0: the call tested on its own, 1: early call that
 * also published its results, 2: late call that reused them, 3: late call that reused them and launched no prepare kernel (the early
 * call had done that work too: it was in order on one stream and directly in front of it). */
uint32_t oxc_debug_shared_tests_mode(const oxc_ctx* ctx);
/* Test hook: which loads the triangle kernels of the context's last oxc_cull_geometry call used for vertex ids / micro indices / positions -- 0: the call
 * ran no triangle stage, 1: `nt` (geometry read once), 2: plain (geometry shared between instances: >= 4 mesh instances per Mesh record, or OXC_TUNE_TRI_LOADS). */
uint32_t oxc_debug_tri_loads_mode(const oxc_ctx* ctx);

/* Harness hook: sizing / scheduling knobs of a context that the measurements and the tests move (the library itself reads no environment
 * variable).  OXC_TUNE_ASYNC_*: resident blocks per CU the persistent kernels of the meshlet / triangle stage take while async_triangles
 * lets the two stages share the machine (0 = no limit, the default).  OXC_TUNE_RASTER_BIG_CAPACITY: entries of oxc_draw_visbuffer's
 * big-triangle / clip queues (default 2^22); only before the context's first draw, which allocates them -- the tests shrink it to reach
 * the overflow paths with a small scene. */
enum { OXC_TUNE_ASYNC_MTEST_BLOCKS_PER_CU = 0, OXC_TUNE_ASYNC_TRI_BLOCKS_PER_CU = 1, OXC_TUNE_RASTER_BIG_CAPACITY = 2,
       OXC_TUNE_TRI_BLOCKS_PER_CU = 3 /* grid cap of the triangle kernels in blocks per CU (default 8 = one resident round) */,
       OXC_TUNE_TRI_LOADS = 7 /* triangle kernels' loads of vertex ids / micro indices / positions: 0 (default) = by the scene -- plain loads when the geometry is shared (>= 4 mesh instances per Mesh record), `nt` when it is unique; 1 = always `nt`; 2 = always plain.  Same outputs either way */,
       OXC_TUNE_MV_EXPAND_ASYNC = 5 /* multi-view batch: blocks per CU of the MeshletInstance expansion on the context's own low-priority stream beside the meshlet stage (default 4); 0: in order on the caller's stream */ };
oxc_status oxc_debug_set_tuning(oxc_ctx* ctx, uint32_t knob, uint32_t value);

/* Measurement aid: counters_dptr != NULL -- the HiZ calls (use_hiz + OXC_CULL_TEST_OCCLUSION) that follow on this context run counting
 * instantiations of their meshlet test, which ADD the number of candidates that reach test_occlusion (cull_meshlets_hiz.slang:53-65:
 * SURVEY 8d's f, four pyramid taps = 16 B each) to 256 u32 counters 256 bytes apart (u32[256 * 64], caller-zeroed; their sum is the
 * count).  Same outputs, slower kernels: not for timed runs.  NULL switches it off again. */
oxc_status oxc_debug_count_occlusion_candidates(oxc_ctx* ctx, void* counters_dptr);

/* Test hook: what the last oxc_draw_visbuffer on this context did with its triangles; synchronises the stream.
 * out4 = {triangles queued for the big path (pixel box beyond 8 x 8), triangles that crossed a clip plane, 64 x 64 tiles handed to
 * the tile list, big-list segments that overflowed (their excess triangles were walked by the setup lane: slow, correct)}. */
oxc_status oxc_debug_raster_stats(oxc_ctx* ctx, uint32_t* host_out4, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* OXCULL_DEBUG_H */
