/*
 * oxcull_oracle.c -- CPU oracle (plain scalar C) for the Oxylus meshlet visibility pipeline.
 *
 * TEST INFRASTRUCTURE ONLY -- see oxcull_oracle.h.  PARITY UNPINNED (the reference has no
 * golden vectors for this path and cannot be built here); this is a restatement of the
 * reference's Slang shaders with the canonical evaluation order of SURVEY.md Appendix A.0.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/Oxylus/src/Render/Shaders unless noted).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * Matrices are column-major (glm): element (row r, col c) = m[c*4+r].  Slang's M[i] is ROW i.
 */
#include "oxcull_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define M(m, r, c) ((m)[(c)*4 + (r)])

/* ------------------------------------------------------------------------------------------
 * helpers: canonical float ops
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
/* Canonical build (the oracle): IEEE binary32, left-to-right, no contraction, true division.
 * -DORC_FAST_ENVELOPE builds liboxcull_oracle_fast.so instead: the SAME algorithm with the rewrites a
 * SLANG_FLOATING_POINT_MODE_FAST / Vulkan driver compiler may legally apply (ResourceCompiler/private/
 * Session.cpp:49-58) -- every a*b+c fused into one rounding, x/y as x * (1/y), normalisation by a reciprocal
 * square root.  It is NOT a second oracle: it exists so the tests and the bench can report how many decisions
 * of a scene such rewrites can flip (the "unpinned gap" of DESIGN.md section 2), next to orc_margin_stats. */
#ifdef ORC_FAST_ENVELOPE
#define MADD(a, b, c) fmaf((a), (b), (c))
#define DIVF(a, b) ((a) * (1.0f / (b)))
#else
#define MADD(a, b, c) ((a) * (b) + (c))
#define DIVF(a, b) ((a) / (b))
#endif
int orc_is_fast_envelope(void) {
#ifdef ORC_FAST_ENVELOPE
  return 1;
#else
  return 0;
#endif
}
static inline float dot3(const float* a, const float* b) { return MADD(a[2], b[2], MADD(a[1], b[1], a[0] * b[0])); }
static inline float len3(const float* a) { return sqrtf(dot3(a, a)); }
static inline float min2(float a, float b) { return fminf(a, b); }
static inline float max2(float a, float b) { return fmaxf(a, b); }

/* float -> u32 as v_cvt_u32_f32 does it: saturating, NaN -> 0. */
static inline uint32_t cvt_u32_sat(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
/* float -> i32 saturating, NaN -> 0 (v_cvt_i32_f32). */
static inline int32_t cvt_i32_sat(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (int32_t)0x80000000;
  return (int32_t)f;
}

/* |a-b| within k ulp of the larger magnitude */
static inline int near_ulp(float a, float b, int k) {
  float m = fmaxf(fabsf(a), fabsf(b));
  if (m == 0.0f) return 1;
  int e;
  frexpf(m, &e);
  float ulp = ldexpf(1.0f, e - 24);
  return fabsf(a - b) <= (float)k * ulp;
}

/* mul(M, v), rows left-to-right (A.0) */
static inline void mul_mv4(const float* m, float x, float y, float z, float w, float* out) {
  for (int i = 0; i < 4; i++) out[i] = MADD(M(m, i, 3), w, MADD(M(m, i, 2), z, MADD(M(m, i, 1), y, M(m, i, 0) * x)));
}
/* mul(M, float4(p, 1.0)): the last product M[i][3]*1.0 is exact, so it is written as an add */
static inline void mul_mp(const float* m, const float* p, float* out) {
  for (int i = 0; i < 4; i++) out[i] = MADD(M(m, i, 2), p[2], MADD(M(m, i, 1), p[1], M(m, i, 0) * p[0])) + M(m, i, 3);
}

/* mul(A, B) -- cull_meshlets.slang:40 `mul(camera.projection_view, transform.world)` */
void orc_mul_mat4(const float* a, const float* b, float* out) {
  float t[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++)
      M(t, r, c) = MADD(M(a, r, 3), M(b, 3, c), MADD(M(a, r, 2), M(b, 2, c), MADD(M(a, r, 1), M(b, 1, c), M(a, r, 0) * M(b, 0, c))));
  memcpy(out, t, sizeof t);
}

/* common/math.slang:193-201 (com::dequantize_half): denormals flush to signed zero,
 * Inf/NaN keep their class. */
float orc_dequantize_half(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000) << 16;
  int32_t em = h & 0x7fff;
  int32_t r = (em + (112 << 10)) << 13;
  r = (em < (1 << 10)) ? 0 : r;
  r += (em >= (31 << 10)) ? (112 << 23) : 0;
  return u2f(s | (uint32_t)r);
}

/* scene.slang:401-435 MeshletBounds::get_* */
void orc_decode_bounds(const orc_meshlet_bounds* b, float* center, float* extent, float* axis, float* cutoff) {
  for (int i = 0; i < 3; i++) {
    center[i] = orc_dequantize_half(b->aabb_center[i]);
    extent[i] = orc_dequantize_half(b->aabb_extent[i]);
  }
  axis[0] = (float)(int32_t)b->cone_axis_xy[0] / 127.0f;
  axis[1] = (float)(int32_t)b->cone_axis_xy[1] / 127.0f;
  axis[2] = (float)(int32_t)b->cone_axis_z / 127.0f;
  *cutoff = (float)(int32_t)b->cone_cutoff / 127.0f;
}

/* cull.slang:49-51 normalize_plane: all four components divided by length(xyz) */
static inline void normalize_plane(const float* p, float* out) {
  float l = len3(p);
  out[0] = DIVF(p[0], l);
  out[1] = DIVF(p[1], l);
  out[2] = DIVF(p[2], l);
  out[3] = DIVF(p[3], l);
}

static void frustum_planes(const float* mvp, float planes[6][4]) {
  float r0[4], r1[4], r2[4], r3[4], t[4];
  for (int c = 0; c < 4; c++) {
    r0[c] = M(mvp, 0, c);
    r1[c] = M(mvp, 1, c);
    r2[c] = M(mvp, 2, c);
    r3[c] = M(mvp, 3, c);
  }
  for (int c = 0; c < 4; c++) t[c] = r3[c] + r0[c]; /* cull.slang:60 left */
  normalize_plane(t, planes[0]);
  for (int c = 0; c < 4; c++) t[c] = r3[c] - r0[c]; /* :62 right */
  normalize_plane(t, planes[1]);
  for (int c = 0; c < 4; c++) t[c] = r3[c] + r1[c]; /* :64 bottom */
  normalize_plane(t, planes[2]);
  for (int c = 0; c < 4; c++) t[c] = r3[c] - r1[c]; /* :66 top */
  normalize_plane(t, planes[3]);
  normalize_plane(r2, planes[4]);                    /* :68 near */
  for (int c = 0; c < 4; c++) t[c] = r3[c] - r2[c]; /* :70 far */
  normalize_plane(t, planes[5]);
}

/* cull.slang:57-84 test_frustum.  near_out (optional): set when a plane comparison is
 * within 4 ulp. */
static int test_frustum_m(const float* mvp, const float* center, const float* extent, int* near_out) {
  float planes[6][4];
  frustum_planes(mvp, planes);
  float h[3] = {extent[0] * 0.5f, extent[1] * 0.5f, extent[2] * 0.5f};
  for (int i = 0; i < 6; i++) {
    float q[3];
    for (int k = 0; k < 3; k++) {
      uint32_t flip = f2u(planes[i][k]) & 0x80000000u;
      q[k] = center[k] + u2f(f2u(h[k]) ^ flip);
    }
    float d = dot3(q, planes[i]);
    float rhs = -planes[i][3];
    if (near_out && near_ulp(d, rhs, 4)) *near_out = 1;
    if (d <= rhs) return 0;
  }
  return 1;
}
int orc_test_frustum(const float* mvp, const float* center, const float* extent) {
  return test_frustum_m(mvp, center, extent, NULL);
}

/* cull.slang:173-175 test_cone (true => culled) */
static int test_cone_m(const float* center, float radius, const float* axis, float cutoff, const float* cam, int* near_out) {
  float d[3] = {center[0] - cam[0], center[1] - cam[1], center[2] - cam[2]};
  float lhs = dot3(d, axis);
  float rhs = MADD(cutoff, len3(d), radius);
  if (near_out && near_ulp(lhs, rhs, 4)) *near_out = 1;
  return lhs >= rhs;
}
int orc_test_cone(const float* center, float radius, const float* axis, float cutoff, const float* cam) {
  return test_cone_m(center, radius, axis, cutoff, cam, NULL);
}

/* scene.slang:292-299 TransformWorld::normal_matrix: cofactor matrix of the upper 3x3.
 * basis = transpose(mat3(world)) => basis[j] (row j of the transpose) = column j of world.
 * result = transpose(mat3(cross(b1,b2), cross(b2,b0), cross(b0,b1))) => column k = k-th cross.
 * out9 is column-major 3x3: element (r,c) = out9[c*3+r]. */
void orc_normal_matrix(const float* w, float* out9) {
  float b[3][3];
  for (int j = 0; j < 3; j++)
    for (int k = 0; k < 3; k++) b[j][k] = M(w, k, j);
  const int a1[3] = {1, 2, 0}, a2[3] = {2, 0, 1};
  for (int k = 0; k < 3; k++) {
    const float* u = b[a1[k]];
    const float* v = b[a2[k]];
    out9[k * 3 + 0] = u[1] * v[2] - v[1] * u[2];
    out9[k * 3 + 1] = u[2] * v[0] - v[2] * u[0];
    out9[k * 3 + 2] = u[0] * v[1] - v[0] * u[1];
  }
}
/* mul(mat3, v) */
static inline void mul_m3v(const float* m9, const float* v, float* out) {
  for (int i = 0; i < 3; i++) out[i] = (m9[0 * 3 + i] * v[0] + m9[1 * 3 + i] * v[1]) + m9[2 * 3 + i] * v[2];
}

/* scene.slang:305-310 to_world_radius -- world[i].xyz is ROW i (A.4) */
float orc_to_world_radius(const float* w, float radius) {
  float r0[3] = {M(w, 0, 0), M(w, 0, 1), M(w, 0, 2)};
  float r1[3] = {M(w, 1, 0), M(w, 1, 1), M(w, 1, 2)};
  float r2[3] = {M(w, 2, 0), M(w, 2, 1), M(w, 2, 2)};
  float sx = len3(r0), sy = len3(r1), sz = len3(r2);
  return radius * max2(sx, max2(sy, sz));
}

/* cull.slang:12-47 project_aabb.  Returns 0 for `none`. out6 = {min.xyz, max.xyz}. */
static int project_aabb_m(const float* mvp, float near_clip, const float* c, const float* e, float* out6, int* near_out) {
  float SX[4], SY[4], SZ[4], P[8][4];
  for (int i = 0; i < 4; i++) {
    SX[i] = M(mvp, i, 0) * e[0]; /* mul(mvp,(ex,0,0,0)): the zero products vanish */
    SY[i] = M(mvp, i, 1) * e[1];
    SZ[i] = M(mvp, i, 2) * e[2];
  }
  float p0[3] = {c[0] - e[0] * 0.5f, c[1] - e[1] * 0.5f, c[2] - e[2] * 0.5f};
  mul_mp(mvp, p0, P[0]);
  for (int i = 0; i < 4; i++) {
    P[1][i] = P[0][i] + SZ[i];
    P[2][i] = P[0][i] + SY[i];
    P[3][i] = P[2][i] + SZ[i];
    P[4][i] = P[0][i] + SX[i];
    P[5][i] = P[4][i] + SZ[i];
    P[6][i] = P[4][i] + SY[i];
    P[7][i] = P[6][i] + SZ[i];
  }
  float depth = P[7][3];
  for (int k = 6; k >= 0; k--) depth = min2(P[k][3], depth);
  if (near_out && near_ulp(depth, near_clip, 4)) *near_out = 1;
  if (depth < near_clip) return 0;
  float vmin[3], vmax[3];
  for (int j = 0; j < 3; j++) {
    float lo = DIVF(P[7][j], P[7][3]);
    float hi = lo;
    for (int k = 6; k >= 0; k--) {
      float d = DIVF(P[k][j], P[k][3]);
      lo = min2(d, lo);
      hi = max2(d, hi);
    }
    vmin[j] = lo;
    vmax[j] = hi;
  }
  out6[0] = vmin[0] * 0.5f + 0.5f;
  out6[1] = vmin[1] * 0.5f + 0.5f;
  out6[2] = vmin[2];
  out6[3] = vmax[0] * 0.5f + 0.5f;
  out6[4] = vmax[1] * 0.5f + 0.5f;
  out6[5] = vmax[2];
  return 1;
}
int orc_project_aabb(const float* mvp, float near_clip, const float* c, const float* e, float* out6) {
  return project_aabb_m(mvp, near_clip, c, e, out6, NULL);
}

static inline uint32_t mip_dim(uint32_t d, uint32_t mip) {
  uint32_t v = d >> mip;
  return v ? v : 1u;
}

/* cull.slang:86-112 sample_level_min_reduction_2x2 */
float orc_sample_level_min_reduction_2x2(const orc_hiz* hiz, float u, float v, uint32_t mip) {
  uint32_t mw = mip_dim(hiz->width, mip), mh = mip_dim(hiz->height, mip);
  float fw = (float)mw, fh = (float)mh;
  int32_t maxx = (int32_t)mw - 1, maxy = (int32_t)mh - 1;
  int32_t bx = cvt_i32_sat(floorf(u * fw - 0.5f));
  int32_t by = cvt_i32_sat(floorf(v * fh - 0.5f));
#define CLAMPI(x, lo, hi) ((x) < (lo) ? (lo) : ((x) > (hi) ? (hi) : (x)))
  int32_t x0 = CLAMPI(bx, 0, maxx), y0 = CLAMPI(by, 0, maxy);
  /* i32 add wraps on the GPU (base + i32x2(1,0)); do it in unsigned to stay defined in C */
  int32_t bx1 = (int32_t)((uint32_t)bx + 1u), by1 = (int32_t)((uint32_t)by + 1u);
  int32_t x1 = CLAMPI(bx1, 0, maxx), y1 = CLAMPI(by1, 0, maxy);
  const float* lvl = hiz->data + hiz->level_offset[mip];
  float p00 = lvl[(size_t)y0 * mw + x0];
  float p10 = lvl[(size_t)y0 * mw + x1];
  float p01 = lvl[(size_t)y1 * mw + x0];
  float p11 = lvl[(size_t)y1 * mw + x1];
  return min2(min2(p00, p10), min2(p01, p11));
}

/* ceil(log2(float(x))) clamped to [0, levels-1], in integers (A.0).  For x > 2^24 the float
 * conversion may round to the next power of two, but levels <= 13 clamps those anyway. */
static inline uint32_t ceil_log2_clamped(uint32_t x, uint32_t levels) {
  uint32_t m = x <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(x - 1u);
  uint32_t top = levels - 1u;
  return m > top ? top : m;
}

static void occlusion_setup(const float* a, const orc_hiz* hiz, uint32_t* mip, float* u, float* v) {
  float sw = (float)hiz->width, sh = (float)hiz->height;
  uint32_t minx = cvt_u32_sat(max2(a[0] * sw, 0.0f));
  uint32_t miny = cvt_u32_sat(max2(a[1] * sh, 0.0f));
  uint32_t maxx = cvt_u32_sat(min2(a[3] * sw, sw - 1.0f));
  uint32_t maxy = cvt_u32_sat(min2(a[4] * sh, sh - 1.0f));
  uint32_t szx = maxx - minx, szy = maxy - miny; /* u32 wrap-around, cull.slang:127 */
  uint32_t ms = szx > szy ? szx : szy;
  *mip = ceil_log2_clamped(ms, hiz->levels);
  *u = (((float)minx + (float)maxx) * 0.5f) / sw;
  *v = (((float)miny + (float)maxy) * 0.5f) / sh;
}

uint32_t orc_occlusion_mip(const float* a, const orc_hiz* hiz) {
  uint32_t mip;
  float u, v;
  occlusion_setup(a, hiz, &mip, &u, &v);
  return mip;
}

/* cull.slang:114-135 test_occlusion (true => occluded) */
static int test_occlusion_m(const float* a, const orc_hiz* hiz, int* near_out) {
  uint32_t mip;
  float u, v;
  occlusion_setup(a, hiz, &mip, &u, &v);
  float d = orc_sample_level_min_reduction_2x2(hiz, u, v, mip);
  float rhs = d - 1e-7f;
  if (near_out && near_ulp(a[5], rhs, 4)) *near_out = 1;
  return a[5] <= rhs;
}
int orc_test_occlusion(const float* a, const orc_hiz* hiz) { return test_occlusion_m(a, hiz, NULL); }

/* cull.slang:169-171: determinant(float3x3(c0.xyw, c1.xyw, c2.xyw)) >= 0.0001, first-row
 * cofactor expansion a(ei-fh) - b(di-fg) + c(dh-eg).  clip3x4: 3 rows of xyzw. */
static int backface_m(const float* cp, int* near_out) {
  float a = cp[0], b = cp[1], c = cp[3];
  float d = cp[4], e = cp[5], f = cp[7];
  float g = cp[8], h = cp[9], i = cp[11];
#ifdef ORC_FAST_ENVELOPE
  float det = fmaf(c, fmaf(d, h, -(e * g)), fmaf(a, fmaf(e, i, -(f * h)), -(b * fmaf(d, i, -(f * g)))));
#else
  float det = (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
#endif
  if (near_out && near_ulp(det, 0.0001f, 4)) *near_out = 1;
  return det >= 0.0001f;
}
int orc_test_triangle_backface(const float* cp) { return backface_m(cp, NULL); }

/* ------------------------------------------------------------------------------------------
 * HiZ -- passes/hiz.slang:42-267, host Passes/CullGeometry.cpp:10-59
 * ---------------------------------------------------------------------------------------- */
void orc_generate_hiz(const float* depth, uint32_t dw, uint32_t dh, orc_hiz* hiz) {
  float* out = (float*)hiz->data;
  uint32_t W = hiz->width, H = hiz->height;
  uint32_t mips = hiz->levels < 13u ? hiz->levels : 13u; /* CullGeometry.cpp:24 */
  /* hiz.slang:32-33 inv_src_extent = 1.0 / f32x2(src_extent), src_extent = HiZ extent
   * (CullGeometry.cpp:29-31). */
  float invx = 1.0f / (float)W, invy = 1.0f / (float)H;
  float* m0 = out + hiz->level_offset[0];
  for (uint32_t y = 0; y < H; y++) {
    /* hiz.slang:92-95 load(): uv = texel*inv + inv; NEAREST clamped sample of the depth
     * image (CullGeometry.cpp:33): texel = floor(uv * depth_dim) clamped to the image. */
    float vv = (float)y * invy + invy;
    int32_t sy = cvt_i32_sat(floorf(vv * (float)dh));
    sy = CLAMPI(sy, 0, (int32_t)dh - 1);
    for (uint32_t x = 0; x < W; x++) {
      float uu = (float)x * invx + invx;
      int32_t sx = cvt_i32_sat(floorf(uu * (float)dw));
      sx = CLAMPI(sx, 0, (int32_t)dw - 1);
      /* transform_z with mat2(1) (CullGeometry.cpp:44): z / 1 == z */
      m0[(size_t)y * W + x] = depth[(size_t)sy * dw + sx];
    }
  }
  /* hiz.slang:77-83 reduce = min over the 2x2 block of the previous mip */
  for (uint32_t k = 1; k < mips; k++) {
    uint32_t pw = mip_dim(W, k - 1), ph = mip_dim(H, k - 1);
    uint32_t cw = mip_dim(W, k), ch = mip_dim(H, k);
    const float* src = out + hiz->level_offset[k - 1];
    float* dst = out + hiz->level_offset[k];
    for (uint32_t y = 0; y < ch; y++)
      for (uint32_t x = 0; x < cw; x++) {
        uint32_t x0 = 2 * x, y0 = 2 * y;
        uint32_t x1 = x0 + 1 < pw ? x0 + 1 : pw - 1, y1 = y0 + 1 < ph ? y0 + 1 : ph - 1;
        float a = src[(size_t)y0 * pw + x0], b = src[(size_t)y0 * pw + x1];
        float c = src[(size_t)y1 * pw + x0], d = src[(size_t)y1 * pw + x1];
        dst[(size_t)y * cw + x] = min2(min2(a, b), min2(c, d));
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * cull_meshes -- passes/cull_meshes.slang:17-85
 * ---------------------------------------------------------------------------------------- */
static inline const float* xform(const float* transforms, uint32_t i) { return transforms + (size_t)i * 16; }

uint32_t orc_cull_meshes(const orc_mesh* meshes, const float* transforms, orc_mesh_instance* mesh_instances,
                         const orc_cull_camera* cam, uint32_t cull_flags, orc_meshlet_instance* out, uint32_t* cmd3) {
  uint32_t total = 0;
  for (uint32_t mi = 0; mi < cam->mesh_instance_count; mi++) {
    orc_mesh_instance* inst = &mesh_instances[mi];
    const orc_mesh* mesh = &meshes[inst->mesh_index];
    const float* world = xform(transforms, inst->transform_index);
    float mvp[16];
    orc_mul_mat4(cam->projection_view, world, mvp);
    uint32_t meshlet_count = 0, lod_index = 0;
    const orc_mesh_lod* lods = (const orc_mesh_lod*)(uintptr_t)mesh->lods;
    if ((cull_flags & ORC_TEST_FRUSTUM) && orc_test_frustum(mvp, mesh->aabb_center, mesh->aabb_extent)) {
      if (cull_flags & ORC_SELECT_LOD) {
        float c4[4], e4[4];
        mul_mp(world, mesh->aabb_center, c4);
        mul_mv4(world, mesh->aabb_extent[0], mesh->aabb_extent[1], mesh->aabb_extent[2], 0.0f, e4);
        float ex = fabsf(e4[0]), ey = fabsf(e4[1]), ez = fabsf(e4[2]);
        float rough = max2(ex, max2(ey, ez));
        float d[3] = {c4[0] - cam->position[0], c4[1] - cam->position[1], c4[2] - cam->position[2]};
        float dist = max2(len3(d) - 0.5f * rough, 0.0f);
        float pixel_size_at_1m = 2.0f / max2(cam->resolution[0], cam->resolution[1]);
        float size_at_1m = rough / dist;
        float px = size_at_1m / pixel_size_at_1m;
        for (uint32_t i = 1; i < mesh->lod_count; i++) {
          float err = px * lods[i].error;
          if (err < cam->acceptable_lod_error)
            lod_index = i;
          else
            break;
        }
      }
      meshlet_count = lods[lod_index].meshlet_count;
    }
    if (meshlet_count > 0) {
      inst->lod_index = lod_index;
      for (uint32_t i = 0; i < meshlet_count; i++) {
        out[total + i].mesh_instance_index = mi;
        out[total + i].meshlet_index = i;
      }
      total += meshlet_count;
    }
  }
  if (cmd3) {
    cmd3[0] = (total + 63u) / 64u; /* cull_meshes.slang:68-70 atomic_max of ceil(total/64) */
    cmd3[1] = 1;
    cmd3[2] = 1;
  }
  return total;
}

/* ------------------------------------------------------------------------------------------
 * per-meshlet common part of cull_meshlets.slang:37-54 / cull_meshlets_hiz.slang:30-59
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float mvp[16];
  float center[3], extent[3];
  int cone_visible;
  uint32_t mask_index;
} meshlet_eval;

static void eval_meshlet(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                         const orc_meshlet_instance* mli, const orc_cull_camera* cam, meshlet_eval* ev, int* near_out) {
  const orc_mesh_instance* inst = &mesh_instances[mli->mesh_instance_index];
  const float* world = xform(transforms, inst->transform_index);
  orc_mul_mat4(cam->projection_view, world, ev->mvp);
  const orc_mesh* mesh = &meshes[inst->mesh_index];
  const orc_mesh_lod* lod = &((const orc_mesh_lod*)(uintptr_t)mesh->lods)[inst->lod_index];
  const orc_meshlet_bounds* b = &((const orc_meshlet_bounds*)(uintptr_t)lod->meshlet_bounds)[mli->meshlet_index];
  float axis[3], cutoff;
  orc_decode_bounds(b, ev->center, ev->extent, axis, &cutoff);
  ev->mask_index = inst->meshlet_instance_visibility_offset + mli->meshlet_index;

  /* cull_meshlets.slang:49-52 */
  float nm[9], na[3];
  orc_normal_matrix(world, nm);
  mul_m3v(nm, axis, na);
  float l = len3(na);
  float cone_axis[3] = {DIVF(na[0], l), DIVF(na[1], l), DIVF(na[2], l)};
  float wc[4];
  mul_mp(world, ev->center, wc);
  float h[3] = {ev->extent[0] * 0.5f, ev->extent[1] * 0.5f, ev->extent[2] * 0.5f};
  float wr = orc_to_world_radius(world, len3(h));
  ev->cone_visible = cutoff >= 1.0f || !test_cone_m(wc, wr, cone_axis, cutoff, cam->position, near_out);
}

uint32_t orc_cull_meshlets(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                           const orc_meshlet_instance* meshlet_instances, uint32_t begin, uint32_t end,
                           const orc_cull_camera* cam, uint32_t* visible_out, orc_margin_stats* stats) {
  uint32_t n = 0;
  for (uint32_t i = begin; i < end; i++) {
    meshlet_eval ev;
    int nr = 0;
    eval_meshlet(meshes, transforms, mesh_instances, &meshlet_instances[i], cam, &ev, stats ? &nr : NULL);
    int vis = ev.cone_visible && test_frustum_m(ev.mvp, ev.center, ev.extent, stats ? &nr : NULL);
    if (vis) visible_out[n++] = i;
    if (stats && nr) stats->meshlets_near_threshold++;
  }
  return n;
}

typedef struct {
  const orc_mesh* meshes;
  const float* transforms;
  const orc_mesh_instance* mesh_instances;
  const orc_meshlet_instance* meshlet_instances;
  const uint32_t* visible;
  const orc_cull_camera* cam;
  uint32_t begin, end, first;
  uint32_t* tmp;
  uint32_t count;
  uint32_t passes;
} mt_job;

static void* mt_meshlets(void* p) {
  mt_job* j = (mt_job*)p;
  for (uint32_t rep = 1; rep < j->passes; rep++) /* extra timing passes; results identical */
    (void)orc_cull_meshlets(j->meshes, j->transforms, j->mesh_instances, j->meshlet_instances, j->begin, j->end, j->cam,
                            j->tmp, NULL);
  j->count = orc_cull_meshlets(j->meshes, j->transforms, j->mesh_instances, j->meshlet_instances, j->begin, j->end, j->cam,
                               j->tmp, NULL);
  return NULL;
}

uint32_t orc_cull_meshlets_mt(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                              const orc_meshlet_instance* meshlet_instances, uint32_t total,
                              const orc_cull_camera* cam, uint32_t* visible_out, uint32_t nthreads) {
  return orc_cull_meshlets_mt_passes(meshes, transforms, mesh_instances, meshlet_instances, total, cam, visible_out, nthreads, 1);
}

uint32_t orc_cull_meshlets_mt_passes(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                     const orc_meshlet_instance* meshlet_instances, uint32_t total,
                                     const orc_cull_camera* cam, uint32_t* visible_out, uint32_t nthreads, uint32_t passes) {
  if (nthreads <= 1 && passes <= 1) return orc_cull_meshlets(meshes, transforms, mesh_instances, meshlet_instances, 0, total, cam, visible_out, NULL);
  if (nthreads < 1) nthreads = 1;
  mt_job* jobs = (mt_job*)calloc(nthreads, sizeof(mt_job));
  pthread_t* th = (pthread_t*)calloc(nthreads, sizeof(pthread_t));
  uint32_t* tmp = (uint32_t*)malloc((size_t)total * 4 + 4);
  for (uint32_t t = 0; t < nthreads; t++) {
    uint32_t b = (uint32_t)((uint64_t)total * t / nthreads), e = (uint32_t)((uint64_t)total * (t + 1) / nthreads);
    jobs[t] = (mt_job){meshes, transforms, mesh_instances, meshlet_instances, NULL, cam, b, e, 0, tmp + b, 0, passes};
    pthread_create(&th[t], NULL, mt_meshlets, &jobs[t]);
  }
  uint32_t n = 0;
  for (uint32_t t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    memcpy(visible_out + n, jobs[t].tmp, (size_t)jobs[t].count * 4);
    n += jobs[t].count;
  }
  free(tmp);
  free(th);
  free(jobs);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * cull_meshlets_hiz -- passes/cull_meshlets_hiz.slang:19-88
 * ---------------------------------------------------------------------------------------- */
uint32_t orc_cull_meshlets_hiz(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                               const orc_meshlet_instance* meshlet_instances, const orc_cull_camera* cam,
                               uint32_t cull_flags, const orc_hiz* hiz, orc_visibility* vis, uint32_t* mask,
                               uint32_t* visible_out, orc_margin_stats* stats) {
  const int late = (cull_flags & ORC_LATE_PASS) != 0;
  const int occl = (cull_flags & ORC_TEST_OCCLUSION) != 0;
  const int occl_or_late = (cull_flags & (ORC_TEST_OCCLUSION | ORC_LATE_PASS)) != 0; /* HAS_FLAG is "any of" */
  uint32_t emitted = 0;
  /* late pass reads early_visible non-atomically after the early pass (:73) */
  const uint32_t early_total = vis->early_visible_meshlet_instances;
  for (uint32_t i = 0; i < vis->total_visible_meshlet_instances; i++) {
    meshlet_eval ev;
    int nr = 0;
    eval_meshlet(meshes, transforms, mesh_instances, &meshlet_instances[i], cam, &ev, stats ? &nr : NULL);
    uint32_t word = 0, bit = 0;
    int was_visible = 1;
    if (occl) { /* :45-51 */
      word = ev.mask_index / 32u;
      bit = 1u << (ev.mask_index - word * 32u);
      was_visible = (mask[word] & bit) != 0;
    }
    int visible = late ? 1 : was_visible;
    visible = visible && ev.cone_visible;
    visible = visible && test_frustum_m(ev.mvp, ev.center, ev.extent, stats ? &nr : NULL);
    if (occl_or_late && visible) { /* :61-65 */
      float sa[6];
      if (project_aabb_m(ev.mvp, cam->near_clip, ev.center, ev.extent, sa, stats ? &nr : NULL))
        visible = !test_occlusion_m(sa, hiz, stats ? &nr : NULL);
    }
    if (visible && (!late || !was_visible)) { /* :67-79 */
      uint32_t index;
      if (!late)
        index = vis->early_visible_meshlet_instances++;
      else
        index = (vis->late_visible_meshlet_instances++) + early_total;
      visible_out[index] = i;
      emitted++;
    }
    if (occl_or_late) { /* :81-87; word/bit are 0 when TestOcclusion is off (reference behaviour) */
      if (visible)
        mask[word] |= bit;
      else
        mask[word] &= ~bit;
    }
    if (stats && nr) stats->meshlets_near_threshold++;
  }
  return emitted;
}

/* ------------------------------------------------------------------------------------------
 * cull_meshlets_hpb -- passes/cull_meshlets_hpb.slang:25-99, cull.slang:137-166,177-179
 * ---------------------------------------------------------------------------------------- */
/* ceil(log2(x)) for a float, from its bits: exact (log2 of a power of two is its exponent),
 * clamped to [0, levels-1]; x <= 0 / NaN -> log2 is -inf/NaN -> the clamp yields 0. */
static inline uint32_t ceil_log2f_clamped(float x, uint32_t levels) {
  if (!(x > 0.0f)) return 0u;
  uint32_t b = f2u(x);
  int32_t e = (int32_t)((b >> 23) & 0xFFu) - 127;
  if (((b >> 23) & 0xFFu) == 0u) return 0u; /* denormal: hugely negative */
  int32_t c = (b & 0x7FFFFFu) ? e + 1 : e;
  if (c < 0) c = 0;
  if (c > (int32_t)levels - 1) c = (int32_t)levels - 1;
  return (uint32_t)c;
}

/* Nearest, clamped SampleLevel of the R8UI array at an integral mip (CullGeometry.cpp:226) */
static inline uint8_t hpb_sample(const orc_hpb* hpb, float u, float v, uint32_t layer, uint32_t mip) {
  uint32_t mw = mip_dim(hpb->width, mip), mh = mip_dim(hpb->height, mip);
  int32_t x = cvt_i32_sat(floorf(u * (float)mw)), y = cvt_i32_sat(floorf(v * (float)mh));
  x = CLAMPI(x, 0, (int32_t)mw - 1);
  y = CLAMPI(y, 0, (int32_t)mh - 1);
  return hpb->data[hpb->level_offset[mip] + (size_t)layer * mw * mh + (size_t)y * mw + (size_t)x];
}
static inline float fractf_(float x) { return x - floorf(x); }

/* cull.slang:137-166 test_vsm_page */
int orc_test_vsm_page(const float* a, const orc_hpb* hpb, uint32_t layer, const int32_t* page_offset) {
  float sw = (float)hpb->width, sh = (float)hpb->height;
  float pox = (float)page_offset[0] / sw, poy = (float)page_offset[1] / sh;
  float box_w = (a[3] - a[0]) * sw, box_h = (a[4] - a[1]) * sh;
  uint32_t mip = ceil_log2f_clamped(max2(box_w, box_h), hpb->levels);
  int tl = hpb_sample(hpb, fractf_(a[0] + pox), fractf_(a[1] + poy), layer, mip) != 0;
  int tr = hpb_sample(hpb, fractf_(a[3] + pox), fractf_(a[1] + poy), layer, mip) != 0;
  int bl = hpb_sample(hpb, fractf_(a[0] + pox), fractf_(a[4] + poy), layer, mip) != 0;
  int br = hpb_sample(hpb, fractf_(a[3] + pox), fractf_(a[4] + poy), layer, mip) != 0;
  return tl | tr | bl | br;
}

uint32_t orc_cull_meshlets_hpb(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                               const orc_meshlet_instance* meshlet_instances, uint32_t total, const orc_cull_camera* cam,
                               const orc_virtual_clipmap* clipmaps, const uint32_t* dirty, uint32_t clipmap_count,
                               const orc_hpb* hpb, uint32_t* visible_out) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < total; i++) {
    const orc_meshlet_instance* mli = &meshlet_instances[i];
    const orc_mesh_instance* inst = &mesh_instances[mli->mesh_instance_index];
    const float* world = xform(transforms, inst->transform_index);
    float mvp[16];
    orc_mul_mat4(cam->projection_view, world, mvp);
    const orc_mesh* mesh = &meshes[inst->mesh_index];
    const orc_mesh_lod* lod = &((const orc_mesh_lod*)(uintptr_t)mesh->lods)[inst->lod_index];
    const orc_meshlet_bounds* b = &((const orc_meshlet_bounds*)(uintptr_t)lod->meshlet_bounds)[mli->meshlet_index];
    float center[3], extent[3], axis[3], cutoff;
    orc_decode_bounds(b, center, extent, axis, &cutoff);
    float nm[9], na[3];
    orc_normal_matrix(world, nm);
    mul_m3v(nm, axis, na);
    float l = len3(na);
    float cone_axis[3] = {na[0] / l, na[1] / l, na[2] / l};
    /* cull.slang:177-179 test_cone_directional: dot(axis, view_dir) >= cutoff */
    int cone_visible = cutoff >= 1.0f || !(dot3(cone_axis, cam->position) >= cutoff);
    if (!(cone_visible && orc_test_frustum(mvp, center, extent))) continue;
    int visible = 0;
    for (uint32_t v = 0; v < clipmap_count; v++) {
      if (dirty[v] == 0u) continue;
      float cmvp[16], sa[6];
      orc_mul_mat4(clipmaps[v].projection_view_mat, world, cmvp);
      if (!orc_test_frustum(cmvp, center, extent)) continue;
      if (orc_project_aabb(cmvp, clipmaps[v].z_near, center, extent, sa))
        visible = orc_test_vsm_page(sa, hpb, v, clipmaps[v].page_offset);
      else
        visible = 1;
      if (visible) break;
    }
    if (visible) visible_out[n++] = i;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------
 * cull_triangles -- passes/cull_triangles.slang:27-90, scene.slang:336-382,478-484,
 * visbuffer.slang:13-14
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t micro_index(const uint32_t* buf, uint32_t byte_offset) { /* scene.slang:336-342 */
  uint32_t pack = buf[byte_offset >> 2];
  return (pack >> ((byte_offset & 3u) * 8u)) & 0xFFu;
}

static uint32_t cull_triangles_impl(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                    const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                    uint32_t count, const orc_cull_camera* cam, uint32_t* out, orc_margin_stats* stats,
                                    uint32_t max_tris, uint32_t corner_bits, int small_triangle_cull) {
  /* corner_bits == 0: SURVEY A.7's pair form (include/oxcull.h wide_triangle_index = 2) -- every index is {u32 meshlet_instance_index, u32 3t+k},
   * two words of `out`; the return value stays the number of indices */
  const int pair = corner_bits == 0u;
  const uint32_t corner_mask = pair ? 0xFFFFFFFFu : (1u << corner_bits) - 1u;
  uint32_t n = 0;
  for (uint32_t s = 0; s < count; s++) {
    uint32_t mli_index = visible[first + s];
    const orc_meshlet_instance* mli = &meshlet_instances[mli_index];
    const orc_mesh_instance* inst = &mesh_instances[mli->mesh_instance_index];
    const orc_mesh* mesh = &meshes[inst->mesh_index];
    const orc_mesh_lod* lod = &((const orc_mesh_lod*)(uintptr_t)mesh->lods)[inst->lod_index];
    const orc_meshlet* ml = &((const orc_meshlet*)(uintptr_t)lod->meshlets)[mli->meshlet_index];
    float mvp[16];
    orc_mul_mat4(cam->projection_view, xform(transforms, inst->transform_index), mvp);
    const uint32_t* micro = (const uint32_t*)(uintptr_t)lod->local_triangle_indices;
    const uint32_t* vidx = (const uint32_t*)(uintptr_t)lod->indirect_vertex_indices;
    const uint16_t* pos = (const uint16_t*)(uintptr_t)mesh->vertex_positions;
    /* one thread per triangle; the kernel has 64 threads (defines.slang:9-11) */
    uint32_t tcount = ml->triangle_count < max_tris ? ml->triangle_count : max_tris;
    for (uint32_t t = 0; t < tcount; t++) {
      float cp[12];
      for (int k = 0; k < 3; k++) {
        uint32_t li = micro_index(micro, ml->local_triangle_index_offset + t * 3u + (uint32_t)k);
        uint32_t vi = vidx[ml->indirect_vertex_index_offset + li];
        float p[3] = {orc_dequantize_half(pos[(size_t)vi * 4 + 0]), orc_dequantize_half(pos[(size_t)vi * 4 + 1]),
                      orc_dequantize_half(pos[(size_t)vi * 4 + 2])};
        mul_mp(mvp, p, &cp[k * 4]);
      }
      int nr = 0;
      int passed = cp[2] >= 0.0f && cp[6] >= 0.0f && cp[10] >= 0.0f;
      passed = passed && !backface_m(cp, stats ? &nr : NULL);
      if (passed && small_triangle_cull) passed = !orc_test_triangle_small(cp, cam->resolution);
      if (stats && nr) stats->triangles_near_threshold++;
      if (passed && pair) {
        for (uint32_t k = 0; k < 3u; k++) {
          out[2u * (size_t)(n + k)] = mli_index;
          out[2u * (size_t)(n + k) + 1u] = t * 3u + k;
        }
        n += 3;
      } else if (passed) {
        uint32_t base = mli_index << corner_bits; /* MESHLET_PRIMITIVE_BITS = 8 in the reference */
        out[n + 0] = base | ((t * 3u + 0u) & corner_mask);
        out[n + 1] = base | ((t * 3u + 1u) & corner_mask);
        out[n + 2] = base | ((t * 3u + 2u) & corner_mask);
        n += 3;
      }
    }
  }
  return n;
}

uint32_t orc_cull_triangles(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                            const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                            uint32_t count, const orc_cull_camera* cam, uint32_t* out, orc_margin_stats* stats) {
  /* one thread per triangle, 64 threads (defines.slang:9-11); 24+8 bit packing (visbuffer.slang:9-14) */
  return cull_triangles_impl(meshes, transforms, mesh_instances, meshlet_instances, visible, first, count, cam, out, stats, 64u, 8u, 0);
}

/* The opt-in small-triangle cull (include/oxcull.h, oxc_cull_geometry_context::small_triangle_cull): no reference
 * behaviour -- the north star names it, the reference's cull_triangles.slang:68-69 stops at clip-z + backface.
 * clip3x4: 3 rows of xyzw.  Returns 1 when the triangle is to be dropped: all three w > 0 and the screen-space
 * bounding box at `resolution` covers no pixel centre.  Canonical arithmetic: true divisions, left to right. */
int orc_test_triangle_small(const float* cp, const float* resolution) {
  float sx[3], sy[3];
  for (int k = 0; k < 3; k++) {
    float w = cp[k * 4 + 3];
    if (!(w > 0.0f)) return 0;
    sx[k] = ((cp[k * 4 + 0] / w) * 0.5f + 0.5f) * resolution[0];
    sy[k] = ((cp[k * 4 + 1] / w) * 0.5f + 0.5f) * resolution[1];
  }
  float lox = min2(min2(sx[0], sx[1]), sx[2]), hix = max2(max2(sx[0], sx[1]), sx[2]);
  float loy = min2(min2(sy[0], sy[1]), sy[2]), hiy = max2(max2(sy[0], sy[1]), sy[2]);
  return floorf(lox + 0.5f) == floorf(hix + 0.5f) || floorf(loy + 0.5f) == floorf(hiy + 0.5f);
}

uint32_t orc_cull_triangles_flags(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                  const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                  uint32_t count, const orc_cull_camera* cam, uint32_t* out, int wide, int small_triangle_cull) {
  /* wide: include/oxcull.h wide_triangle_index -- 0 packed 24 + 8, 1 packed 23 + 9, 2 {id, corner} pairs (`out` holds two words per index) */
  return cull_triangles_impl(meshes, transforms, mesh_instances, meshlet_instances, visible, first, count, cam, out, NULL, wide ? 128u : 64u,
                             wide == 2 ? 0u : wide ? 9u : 8u, small_triangle_cull);
}

/* Which triangles of the visible slots sit in the BOUNDARY SET of cull_triangles' two tests: the ones whose decision a
 * different (legal, fast-math) evaluation order can flip.  The backface determinant is a sum of six triple products that
 * mostly cancel (tiny or distant triangles: |det| << the products), so "within a few ulp of the threshold" says nothing;
 * the forward error of ANY evaluation order is bounded by a small multiple of eps * (sum of the |products|), and that is
 * the criterion here: |det - 0.0001| <= 4 * 2^-24 * sum|products| (measured: a factor 2 already covers every flip of the fast-math envelope build; the factor covers the roundings of the clip coordinates
 * themselves, which enter each product).  Same for the three clip.z >= 0 tests against sum|terms| of the row-2 dot product.
 * flags64[s * 64 + t] = 1 for a boundary triangle, 0 otherwise (t < min(triangle_count, 64)). */
void orc_triangle_boundary_flags(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                 const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                 uint32_t count, const orc_cull_camera* cam, uint8_t* flags64) {
  const float eps = 4.0f * 5.9604644775390625e-08f;
  for (uint32_t s = 0; s < count; s++) {
    uint32_t mli_index = visible[first + s];
    const orc_meshlet_instance* mli = &meshlet_instances[mli_index];
    const orc_mesh_instance* inst = &mesh_instances[mli->mesh_instance_index];
    const orc_mesh* mesh = &meshes[inst->mesh_index];
    const orc_mesh_lod* lod = &((const orc_mesh_lod*)(uintptr_t)mesh->lods)[inst->lod_index];
    const orc_meshlet* ml = &((const orc_meshlet*)(uintptr_t)lod->meshlets)[mli->meshlet_index];
    float mvp[16];
    orc_mul_mat4(cam->projection_view, xform(transforms, inst->transform_index), mvp);
    const uint32_t* micro = (const uint32_t*)(uintptr_t)lod->local_triangle_indices;
    const uint32_t* vidx = (const uint32_t*)(uintptr_t)lod->indirect_vertex_indices;
    const uint16_t* pos = (const uint16_t*)(uintptr_t)mesh->vertex_positions;
    uint32_t tcount = ml->triangle_count < 64u ? ml->triangle_count : 64u;
    for (uint32_t t = 0; t < 64u; t++) {
      uint8_t flag = 0;
      if (t < tcount) {
        float cp[12];
        for (int k = 0; k < 3; k++) {
          uint32_t li = micro_index(micro, ml->local_triangle_index_offset + t * 3u + (uint32_t)k);
          uint32_t vi = vidx[ml->indirect_vertex_index_offset + li];
          float p[3] = {orc_dequantize_half(pos[(size_t)vi * 4 + 0]), orc_dequantize_half(pos[(size_t)vi * 4 + 1]),
                        orc_dequantize_half(pos[(size_t)vi * 4 + 2])};
          mul_mp(mvp, p, &cp[k * 4]);
          float zmag = (fabsf(M(mvp, 2, 0) * p[0]) + fabsf(M(mvp, 2, 1) * p[1])) + (fabsf(M(mvp, 2, 2) * p[2]) + fabsf(M(mvp, 2, 3)));
          if (fabsf(cp[k * 4 + 2]) <= eps * zmag) flag = 1;
        }
        float a = cp[0], b = cp[1], c = cp[3], d = cp[4], e = cp[5], f = cp[7], g = cp[8], h = cp[9], i = cp[11];
        float det = (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
        float mag = (fabsf(a) * (fabsf(e * i) + fabsf(f * h)) + fabsf(b) * (fabsf(d * i) + fabsf(f * g))) + fabsf(c) * (fabsf(d * h) + fabsf(e * g));
        if (fabsf(det - 0.0001f) <= eps * mag) flag = 1;
      }
      flags64[(size_t)s * 64 + t] = flag;
    }
  }
}

uint32_t orc_cull_triangles_wide(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                 const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                 uint32_t count, const orc_cull_camera* cam, uint32_t* out) {
  return cull_triangles_impl(meshes, transforms, mesh_instances, meshlet_instances, visible, first, count, cam, out, NULL, 128u, 9u, 0);
}

static void* mt_triangles(void* p) {
  mt_job* j = (mt_job*)p;
  j->count = orc_cull_triangles(j->meshes, j->transforms, j->mesh_instances, j->meshlet_instances, j->visible,
                                j->first + j->begin, j->end - j->begin, j->cam, j->tmp, NULL);
  return NULL;
}

uint32_t orc_cull_triangles_mt(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                               const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                               uint32_t count, const orc_cull_camera* cam, uint32_t* out, uint32_t nthreads) {
  if (nthreads <= 1) return orc_cull_triangles(meshes, transforms, mesh_instances, meshlet_instances, visible, first, count, cam, out, NULL);
  mt_job* jobs = (mt_job*)calloc(nthreads, sizeof(mt_job));
  pthread_t* th = (pthread_t*)calloc(nthreads, sizeof(pthread_t));
  uint32_t* tmp = (uint32_t*)malloc((size_t)count * 192 * 4 + 4);
  for (uint32_t t = 0; t < nthreads; t++) {
    uint32_t b = (uint32_t)((uint64_t)count * t / nthreads), e = (uint32_t)((uint64_t)count * (t + 1) / nthreads);
    jobs[t] = (mt_job){meshes, transforms, mesh_instances, meshlet_instances, visible, cam, b, e, first, tmp + (size_t)b * 192, 0, 1};
    pthread_create(&th[t], NULL, mt_triangles, &jobs[t]);
  }
  uint32_t n = 0;
  for (uint32_t t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    memcpy(out + n, jobs[t].tmp, (size_t)jobs[t].count * 4);
    n += jobs[t].count;
  }
  free(tmp);
  free(th);
  free(jobs);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Config 1 harness: ECS transform update + host AABB frustum test.
 * Scene.cpp:1690-1711 (T*R*S chained through parents), BoundingVolume.cpp:32-53
 * (AABB::transform), :72-88 (is_on_or_forward_plane / is_on_frustum).
 * trs10 per entity: translation xyz, quaternion wxyz, scale xyz.  parent[i] < i or -1.
 * planes24: 6 planes {nx,ny,nz,distance} with unit normals (Frustum.hpp:7-18).
 * ---------------------------------------------------------------------------------------- */
static void trs_to_mat(const float* t, float* m) {
  float qw = t[3], qx = t[4], qy = t[5], qz = t[6];
  /* glm::mat4_cast */
  float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz, qxz = qx * qz, qxy = qx * qy, qyz = qy * qz, qwx = qw * qx,
        qwy = qw * qy, qwz = qw * qz;
  float r[9];
  r[0] = 1.0f - 2.0f * (qyy + qzz);
  r[1] = 2.0f * (qxy + qwz);
  r[2] = 2.0f * (qxz - qwy);
  r[3] = 2.0f * (qxy - qwz);
  r[4] = 1.0f - 2.0f * (qxx + qzz);
  r[5] = 2.0f * (qyz + qwx);
  r[6] = 2.0f * (qxz + qwy);
  r[7] = 2.0f * (qyz - qwx);
  r[8] = 1.0f - 2.0f * (qxx + qyy);
  for (int c = 0; c < 3; c++) {
    for (int rr = 0; rr < 3; rr++) M(m, rr, c) = r[c * 3 + rr] * t[7 + c];
    M(m, 3, c) = 0.0f;
  }
  M(m, 0, 3) = t[0];
  M(m, 1, 3) = t[1];
  M(m, 2, 3) = t[2];
  M(m, 3, 3) = 1.0f;
}

uint32_t orc_entities_update_and_cull(uint32_t n, const float* trs10, const int32_t* parent, const float* aabb6,
                                      const float* planes24, float* world_out, uint8_t* visible_out) {
  uint32_t nvis = 0;
  for (uint32_t i = 0; i < n; i++) {
    float local[16];
    trs_to_mat(trs10 + (size_t)i * 10, local);
    float* w = world_out + (size_t)i * 16;
    if (parent[i] >= 0)
      orc_mul_mat4(world_out + (size_t)parent[i] * 16, local, w);
    else
      memcpy(w, local, sizeof local);
    const float* mn = aabb6 + (size_t)i * 6;
    const float* mx = mn + 3;
    float c[3] = {(mx[0] + mn[0]) * 0.5f, (mx[1] + mn[1]) * 0.5f, (mx[2] + mn[2]) * 0.5f};
    float e[3] = {(mx[0] - mn[0]) * 0.5f, (mx[1] - mn[1]) * 0.5f, (mx[2] - mn[2]) * 0.5f};
    float nc[4];
    mul_mp(w, c, nc);
    float ne[3];
    for (int r = 0; r < 3; r++)
      ne[r] = (fabsf(M(w, r, 0)) * e[0] + fabsf(M(w, r, 1)) * e[1]) + fabsf(M(w, r, 2)) * e[2];
    int vis = 1;
    for (int p = 0; p < 6 && vis; p++) {
      const float* pl = planes24 + p * 4;
      float rr = (ne[0] * fabsf(pl[0]) + ne[1] * fabsf(pl[1])) + ne[2] * fabsf(pl[2]);
      float dist = dot3(pl, nc) - pl[3];
      vis = -rr <= dist;
    }
    visible_out[i] = (uint8_t)vis;
    nvis += (uint32_t)vis;
  }
  return nvis;
}

/* `passes` repetitions of the whole update (timing loop of bench.py --workload config1; results identical). */
uint32_t orc_entities_update_and_cull_passes(uint32_t n, const float* trs10, const int32_t* parent, const float* aabb6,
                                             const float* planes24, float* world_out, uint8_t* visible_out, uint32_t passes) {
  uint32_t nvis = 0;
  for (uint32_t p = 0; p < (passes ? passes : 1u); p++) nvis = orc_entities_update_and_cull(n, trs10, parent, aabb6, planes24, world_out, visible_out);
  return nvis;
}

/* ------------------------------------------------------------------------------------------
 * meshlet bounds producer.  See the header for provenance (meshoptimizer v1.2 is a
 * third-party dependency absent from /root/reference; its published algorithm is restated).
 * ------------------------------------------------------------------------------------------ */
/* meshoptimizer.h meshopt_quantizeHalf: round to nearest (ties away from zero in magnitude), flush
 * results below 2^-14 to zero, >= 65536 to infinity, every NaN to a quiet NaN. */
uint16_t orc_quantize_half(float v) {
  uint32_t ui = f2u(v);
  int s = (int)((ui >> 16) & 0x8000u);
  int em = (int)(ui & 0x7fffffffu);
  int h = (em - (112 << 23) + (1 << 12)) >> 13;
  h = (em < (113 << 23)) ? 0 : h;
  h = (em >= (143 << 23)) ? 0x7c00 : h;
  h = (em > (255 << 23)) ? 0x7e00 : h;
  return (uint16_t)(s | h);
}

/* meshoptimizer.h meshopt_quantizeSnorm */
int orc_quantize_snorm(float v, int bits) {
  const float scale = (float)((1 << (bits - 1)) - 1);
  float round = (v >= 0 ? 0.5f : -0.5f);
  v = (v >= -1) ? v : -1;
  v = (v <= +1) ? v : +1;
  return (int)(v * scale + round);
}

/* AssetManager_GLTF.cpp:570-588 */
void orc_quantize_vertex_streams(const float* positions, const float* normals, const float* texcoords, uint32_t vertex_count,
                                 uint16_t* out_qpos, uint32_t* out_qnrm, uint16_t* out_quv) {
  for (uint32_t v = 0; v < vertex_count; v++) {
    if (positions) {
      for (int k = 0; k < 3; k++) out_qpos[4 * v + k] = orc_quantize_half(positions[3 * v + k]);
      out_qpos[4 * v + 3] = 0; /* glm::u16vec4 value-initialised by resize(), only xyz assigned */
    }
    if (normals)
      out_qnrm[v] = ((uint32_t)(orc_quantize_snorm(normals[3 * v + 0], 10) + 511) << 20) |
                    ((uint32_t)(orc_quantize_snorm(normals[3 * v + 1], 10) + 511) << 10) |
                    (uint32_t)(orc_quantize_snorm(normals[3 * v + 2], 10) + 511);
    if (texcoords)
      for (int k = 0; k < 2; k++) out_quv[2 * v + k] = orc_quantize_half(texcoords[2 * v + k]);
  }
}

/* clusterizer.cpp computeBoundingSphere with all radii 0 and axis_count = 3 (the call that fits the normal
 * cone): seed with the most distant pair among the per-axis extrema, then one sweep growing the sphere. */
static void bounding_sphere_axes3(float result[4], const float (*points)[3], uint32_t count) {
  uint32_t pmin[3] = {0, 0, 0}, pmax[3] = {0, 0, 0};
  float tmin[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
  float tmax[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
  for (uint32_t i = 0; i < count; i++) {
    for (int axis = 0; axis < 3; axis++) {
      /* tp = ax[0]*p[0] + ax[1]*p[1] + ax[2]*p[2] with a unit axis: the other products are exact zeros */
      float tp = points[i][axis];
      pmin[axis] = (tp < tmin[axis]) ? i : pmin[axis];
      pmax[axis] = (tp > tmax[axis]) ? i : pmax[axis];
      tmin[axis] = (tp < tmin[axis]) ? tp : tmin[axis];
      tmax[axis] = (tp > tmax[axis]) ? tp : tmax[axis];
    }
  }
  int paxis = 0;
  float paxisdr = 0.0f;
  for (int axis = 0; axis < 3; axis++) {
    const float* p1 = points[pmin[axis]];
    const float* p2 = points[pmax[axis]];
    float d2 = ((p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1])) + (p2[2] - p1[2]) * (p2[2] - p1[2]);
    float dr = sqrtf(d2);
    if (dr > paxisdr) {
      paxisdr = dr;
      paxis = axis;
    }
  }
  const float* p1 = points[pmin[paxis]];
  const float* p2 = points[pmax[paxis]];
  float paxisd = sqrtf(((p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1])) + (p2[2] - p1[2]) * (p2[2] - p1[2]));
  float paxisk = paxisd > 0.0f ? paxisd / (2.0f * paxisd) : 0.0f;
  float center[3] = {p1[0] + (p2[0] - p1[0]) * paxisk, p1[1] + (p2[1] - p1[1]) * paxisk, p1[2] + (p2[2] - p1[2]) * paxisk};
  float radius = paxisdr / 2.0f;
  for (uint32_t i = 0; i < count; i++) {
    const float* p = points[i];
    float d2 = ((p[0] - center[0]) * (p[0] - center[0]) + (p[1] - center[1]) * (p[1] - center[1])) + (p[2] - center[2]) * (p[2] - center[2]);
    float d = sqrtf(d2);
    if (d > radius) {
      float k = d > 0.0f ? (d - radius) / (2.0f * d) : 0.0f;
      center[0] += k * (p[0] - center[0]);
      center[1] += k * (p[1] - center[1]);
      center[2] += k * (p[2] - center[2]);
      radius = (radius + d) / 2.0f;
    }
  }
  result[0] = center[0];
  result[1] = center[1];
  result[2] = center[2];
  result[3] = radius;
}

void orc_build_meshlet_bounds(const float* positions, uint32_t vertex_count, const orc_meshlet* meshlets, uint32_t meshlet_count,
                              const uint32_t* indirect_vertex_indices, const uint8_t* local_triangle_indices,
                              orc_meshlet_bounds* out_bounds, float* out_mesh6, uint16_t* out_qpos) {
  /* AssetManager_GLTF.cpp:573-578 */
  if (out_qpos) {
    for (uint32_t v = 0; v < vertex_count; v++) {
      out_qpos[v * 4 + 0] = orc_quantize_half(positions[v * 3 + 0]);
      out_qpos[v * 4 + 1] = orc_quantize_half(positions[v * 3 + 1]);
      out_qpos[v * 4 + 2] = orc_quantize_half(positions[v * 3 + 2]);
      out_qpos[v * 4 + 3] = 0;
    }
  }
  const float fmax_ = 3.402823466e+38f;
  float mesh_min[3] = {fmax_, fmax_, fmax_}, mesh_max[3] = {-fmax_, -fmax_, -fmax_};
  for (uint32_t m = 0; m < meshlet_count; m++) {
    const orc_meshlet* ml = &meshlets[m];
    const uint32_t* mv = indirect_vertex_indices + ml->indirect_vertex_index_offset;
    const uint8_t* mt = local_triangle_indices + ml->local_triangle_index_offset;
    /* AssetManager_GLTF.cpp:690-706: AABB over every referenced corner (glm::min(a,b) = b < a ? b : a) */
    float bb_min[3] = {fmax_, fmax_, fmax_}, bb_max[3] = {-fmax_, -fmax_, -fmax_};
    for (uint32_t i = 0; i < ml->triangle_count * 3u; i++) {
      const float* p = positions + (size_t)mv[mt[i]] * 3;
      for (int k = 0; k < 3; k++) {
        bb_min[k] = (p[k] < bb_min[k]) ? p[k] : bb_min[k];
        bb_max[k] = (bb_max[k] < p[k]) ? p[k] : bb_max[k];
      }
    }
    /* meshopt_computeMeshletBounds -> meshopt_computeClusterBounds */
    float normals[256][3];
    uint32_t triangles = 0;
    uint32_t tcount = ml->triangle_count < 256u ? ml->triangle_count : 256u;
    for (uint32_t t = 0; t < tcount; t++) {
      const float* p0 = positions + (size_t)mv[mt[t * 3 + 0]] * 3;
      const float* p1 = positions + (size_t)mv[mt[t * 3 + 1]] * 3;
      const float* p2 = positions + (size_t)mv[mt[t * 3 + 2]] * 3;
      float p10[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
      float p20[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
      float nx = p10[1] * p20[2] - p10[2] * p20[1];
      float ny = p10[2] * p20[0] - p10[0] * p20[2];
      float nz = p10[0] * p20[1] - p10[1] * p20[0];
      float area = sqrtf((nx * nx + ny * ny) + nz * nz);
      if (area == 0.0f) continue; /* degenerate triangles are left out */
      normals[triangles][0] = nx / area;
      normals[triangles][1] = ny / area;
      normals[triangles][2] = nz / area;
      triangles++;
    }
    int8_t axis_s8[3] = {0, 0, 0};
    int8_t cutoff_s8 = 0; /* no valid triangle: cone data stays 0 */
    if (triangles > 0) {
      float nsphere[4];
      bounding_sphere_axes3(nsphere, normals, triangles);
      float axis[3] = {nsphere[0], nsphere[1], nsphere[2]};
      float axislength = sqrtf((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
      float invaxislength = axislength == 0.0f ? 0.0f : 1.0f / axislength;
      axis[0] *= invaxislength;
      axis[1] *= invaxislength;
      axis[2] *= invaxislength;
      float mindp = 1.0f;
      for (uint32_t i = 0; i < triangles; i++) {
        float dp = (normals[i][0] * axis[0] + normals[i][1] * axis[1]) + normals[i][2] * axis[2];
        mindp = (dp < mindp) ? dp : mindp;
      }
      if (mindp <= 0.1f) {
        cutoff_s8 = 127; /* cone wider than ~168 degrees: never culls; the axis stays 0 */
      } else {
        float cone_cutoff = sqrtf(1.0f - mindp * mindp);
        for (int k = 0; k < 3; k++) axis_s8[k] = (int8_t)orc_quantize_snorm(axis[k], 8);
        float e0 = fabsf((float)axis_s8[0] / 127.0f - axis[0]);
        float e1 = fabsf((float)axis_s8[1] / 127.0f - axis[1]);
        float e2 = fabsf((float)axis_s8[2] / 127.0f - axis[2]);
        int c = (int)(127.0f * (((cone_cutoff + e0) + e1) + e2) + 1.0f); /* rounded up, not to nearest */
        cutoff_s8 = (c > 127) ? 127 : (int8_t)c;
      }
    }
    /* AssetManager_GLTF.cpp:717-735 */
    orc_meshlet_bounds* b = &out_bounds[m];
    for (int k = 0; k < 3; k++) {
      float center = (bb_max[k] + bb_min[k]) * 0.5f;
      float extent = bb_max[k] - bb_min[k];
      b->aabb_center[k] = orc_quantize_half(center);
      b->aabb_extent[k] = orc_quantize_half(extent);
      mesh_min[k] = (bb_min[k] < mesh_min[k]) ? bb_min[k] : mesh_min[k];
      mesh_max[k] = (mesh_max[k] < bb_max[k]) ? bb_max[k] : mesh_max[k];
    }
    b->cone_axis_xy[0] = axis_s8[0];
    b->cone_axis_xy[1] = axis_s8[1];
    b->cone_axis_z = axis_s8[2];
    b->cone_cutoff = cutoff_s8;
  }
  /* AssetManager_GLTF.cpp:741-744 */
  for (int k = 0; k < 3; k++) {
    out_mesh6[k] = (mesh_max[k] + mesh_min[k]) * 0.5f;
    out_mesh6[3 + k] = mesh_max[k] - mesh_min[k];
  }
}

/* ------------------------------------------------------------------------------------------
 * HPB producer (passes/rmvsm_downsample_hpb.slang:10-33, Shadowmaps.cpp:331-366).
 * ------------------------------------------------------------------------------------------ */
void orc_generate_hpb(const uint32_t* page_table, orc_hpb* hpb) {
  uint8_t* base = (uint8_t*)(uintptr_t)hpb->data;
  for (uint32_t lvl = 0; lvl < hpb->levels; lvl++) {
    uint32_t w = hpb->width >> lvl, h = hpb->height >> lvl; /* Shadowmaps.cpp:342-346 */
    w = w ? w : 1u;
    h = h ? h : 1u;
    uint8_t* dst = base + hpb->level_offset[lvl];
    if (lvl == 0) { /* IS_FIRST_PASS: page.is_visible() && page.is_backed() && page.is_dirty() */
      for (size_t i = 0; i < (size_t)hpb->layers * w * h; i++) {
        uint32_t page = page_table[i];
        dst[i] = (uint8_t)(((page & 1u) != 0u) && ((page & 4u) != 0u) && ((page & 2u) != 0u));
      }
      continue;
    }
    uint32_t sw = hpb->width >> (lvl - 1), sh = hpb->height >> (lvl - 1);
    sw = sw ? sw : 1u;
    sh = sh ? sh : 1u;
    const uint8_t* src = base + hpb->level_offset[lvl - 1];
    for (uint32_t z = 0; z < hpb->layers; z++)
      for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
          uint32_t acc = 0; /* tl | tr | bl | br, out-of-range Loads return 0 */
          for (uint32_t dy = 0; dy < 2; dy++)
            for (uint32_t dx = 0; dx < 2; dx++) {
              uint32_t sx = x * 2 + dx, sy = y * 2 + dy;
              if (sx < sw && sy < sh) acc |= src[((size_t)z * sh + sy) * sw + sx];
            }
          dst[((size_t)z * h + y) * w + x] = (uint8_t)(acc == 1u);
        }
  }
}

/* ------------------------------------------------------------------------------------------
 * terrain patch cull, passes/terrain_cull.slang:17-83 (TerrainData helpers scene.slang:648-660).
 * ------------------------------------------------------------------------------------------ */
uint32_t orc_cull_terrain(const float* world_min2, const float* world_size2, uint32_t pcx, uint32_t pcy, float base_height, float height_scale,
                          const float* patch_minmax, const orc_cull_camera* cam, uint32_t cull_flags, const orc_hiz* hiz, uint32_t* mask,
                          uint32_t* out_visible) {
  const uint32_t total = pcx * pcy;
  const int late = (cull_flags & ORC_LATE_PASS) != 0;
  const int occl_or_late = (cull_flags & (ORC_TEST_OCCLUSION | ORC_LATE_PASS)) != 0;
  uint32_t n = 0;
  for (uint32_t patch_index = 0; patch_index < total; patch_index++) {
    uint32_t px = patch_index % pcx, py = patch_index / pcx;
    /* patch_corner: world_min + (f32x2(patch + corner) / f32x2(patch_count)) * world_size */
    float g0x = (float)(px + 0u) / (float)pcx, g0y = (float)(py + 0u) / (float)pcy;
    float g1x = (float)(px + 1u) / (float)pcx, g1y = (float)(py + 1u) / (float)pcy;
    float cminx = world_min2[0] + g0x * world_size2[0], cminy = world_min2[1] + g0y * world_size2[1];
    float cmaxx = world_min2[0] + g1x * world_size2[0], cmaxy = world_min2[1] + g1y * world_size2[1];
    float bx = patch_minmax[(size_t)patch_index * 2 + 0], by = patch_minmax[(size_t)patch_index * 2 + 1];
    float center[3] = {(cminx + cmaxx) * 0.5f, base_height + ((bx + by) * 0.5f) * height_scale, (cminy + cmaxy) * 0.5f};
    float hy = height_scale * (by - bx);
    float extent[3] = {cmaxx - cminx, hy > 1e-3f ? hy : 1e-3f, cmaxy - cminy}; /* max(a, 1e-3) */
    uint32_t word = patch_index / 32u, bit = 1u << (patch_index % 32u);
    int was_visible = (mask[word] & bit) != 0u;
    int visible = late ? 1 : was_visible;
    if (cull_flags & ORC_TEST_FRUSTUM) visible = visible && orc_test_frustum(cam->projection_view, center, extent);
    if (occl_or_late && visible) {
      float sa[6];
      if (orc_project_aabb(cam->projection_view, cam->near_clip, center, extent, sa)) visible = !orc_test_occlusion(sa, hiz);
    }
    int emit = visible && (!late || !was_visible);
    if (occl_or_late) {
      if (visible)
        mask[word] |= bit;
      else
        mask[word] &= ~bit;
    }
    if (emit) out_visible[n++] = patch_index;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------
 * consumer of the indirect draw (see the header for the stated raster rules).
 * ------------------------------------------------------------------------------------------ */
static int64_t edge_fn(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py) {
  return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
/* top-left rule for triangles oriented with positive area: an edge owns its pixels when it goes down,
 * or is horizontal going left (complementary for the neighbour that shares the edge reversed) */
static int edge_inclusive(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
  int64_t dx = bx - ax, dy = by - ay;
  return dy > 0 || (dy == 0 && dx < 0);
}

/* clip planes of the stated rules: w >= 2^-10 and the guard band |x|, |y| <= 64 w */
#define ORC_CLIP_WMIN 0.0009765625f
#define ORC_CLIP_GUARD 64.0f
static float clip_distance(const float* v, int plane) {
  switch (plane) {
    case 0: return v[3] - ORC_CLIP_WMIN;
    case 1: return ORC_CLIP_GUARD * v[3] - v[0];
    case 2: return ORC_CLIP_GUARD * v[3] + v[0];
    case 3: return ORC_CLIP_GUARD * v[3] - v[1];
    default: return ORC_CLIP_GUARD * v[3] + v[1];
  }
}

/* setup + coverage + depth for one (possibly clipped) triangle given in clip coordinates */
static void raster_clip_triangle(const float* c0, const float* c1, const float* c2, uint32_t vis_out, uint32_t W, uint32_t H, uint64_t* visdepth) {
  const float* cl[3] = {c0, c1, c2};
  int64_t X[3], Y[3];
  float z[3];
  for (int k = 0; k < 3; k++) {
    const float* clip = cl[k];
    if (!(clip[3] > 0.0f)) return;
    float sx = ((clip[0] / clip[3]) * 0.5f + 0.5f) * (float)W;
    float sy = ((clip[1] / clip[3]) * 0.5f + 0.5f) * (float)H;
    z[k] = clip[2] / clip[3];
    if (!(fabsf(sx) <= 1048576.0f) || !(fabsf(sy) <= 1048576.0f)) return;
    X[k] = (int64_t)floorf(sx * 256.0f + 0.5f);
    Y[k] = (int64_t)floorf(sy * 256.0f + 0.5f);
  }
  int64_t area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
  if (area >= 0) return; /* cullMode eBack: det(xyw) > 0 <=> positive area (cull.slang:169-171); 0 = no coverage */
  /* orient positively: swap corners 1 and 2 */
  int64_t t = X[1]; X[1] = X[2]; X[2] = t;
  t = Y[1]; Y[1] = Y[2]; Y[2] = t;
  float tz = z[1]; z[1] = z[2]; z[2] = tz;
  area = -area;
  int64_t minx = X[0] < X[1] ? X[0] : X[1], maxx = X[0] > X[1] ? X[0] : X[1];
  int64_t miny = Y[0] < Y[1] ? Y[0] : Y[1], maxy = Y[0] > Y[1] ? Y[0] : Y[1];
  minx = minx < X[2] ? minx : X[2]; maxx = maxx > X[2] ? maxx : X[2];
  miny = miny < Y[2] ? miny : Y[2]; maxy = maxy > Y[2] ? maxy : Y[2];
  /* pixel (px,py) has its centre at (256 px + 128, 256 py + 128) */
  int64_t px0 = (minx - 128 + 255) >> 8, px1 = (maxx - 128) >> 8;
  int64_t py0 = (miny - 128 + 255) >> 8, py1 = (maxy - 128) >> 8;
  if (px0 < 0) px0 = 0;
  if (py0 < 0) py0 = 0;
  if (px1 > (int64_t)W - 1) px1 = (int64_t)W - 1;
  if (py1 > (int64_t)H - 1) py1 = (int64_t)H - 1;
  const int64_t b0 = edge_inclusive(X[1], Y[1], X[2], Y[2]) ? 0 : -1;
  const int64_t b1 = edge_inclusive(X[2], Y[2], X[0], Y[0]) ? 0 : -1;
  const int64_t b2 = edge_inclusive(X[0], Y[0], X[1], Y[1]) ? 0 : -1;
  const double inv_area = 1.0 / (double)area; /* one reciprocal per triangle */
  for (int64_t py = py0; py <= py1; py++)
    for (int64_t px = px0; px <= px1; px++) {
      int64_t cxp = px * 256 + 128, cyp = py * 256 + 128;
      int64_t e0 = edge_fn(X[1], Y[1], X[2], Y[2], cxp, cyp); /* weight of corner 0 */
      int64_t e1 = edge_fn(X[2], Y[2], X[0], Y[0], cxp, cyp);
      int64_t e2 = edge_fn(X[0], Y[0], X[1], Y[1], cxp, cyp);
      if (e0 + b0 < 0 || e1 + b1 < 0 || e2 + b2 < 0) continue;
      double zd = (((double)e0 * (double)z[0] + (double)e1 * (double)z[1]) + (double)e2 * (double)z[2]) * inv_area;
      float zf = (float)zd;
      if (!(zf > 0.0f) || zf > 1.0f) continue;
      uint64_t packed = ((uint64_t)f2u(zf) << 32) | vis_out;
      uint64_t* dst = &visdepth[(size_t)py * W + (size_t)px];
      if (packed > *dst) *dst = packed;
    }
}

static uint32_t g_draw_clipped; /* triangles of the last orc_draw_visbuffer that crossed a clip plane (tests: "the clipper ran") */
uint32_t orc_draw_clipped_count(void) { return g_draw_clipped; }

void orc_draw_visbuffer(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                        const orc_meshlet_instance* meshlet_instances, const uint32_t* indices, uint32_t index_count, const float* pv, uint32_t W,
                        uint32_t H, uint32_t corner_bits, uint64_t* visdepth) {
  const int pair = corner_bits == 0u; /* {id, corner} pairs: `indices` holds two words per index, index_count counts indices */
  const uint32_t corner_mask = pair ? 0xFFFFFFFFu : (1u << corner_bits) - 1u;
  g_draw_clipped = 0;
  for (uint32_t i = 0; i + 2 < index_count; i += 3) {
    float poly[2][9][4];
    uint32_t vis_out = 0;
    for (int k = 0; k < 3; k++) {
      /* vs_main, visbuffer_encode.slang:24-49 */
      uint32_t data = pair ? 0u : indices[i + k];
      uint32_t mli_index = pair ? indices[2u * (size_t)(i + k)] : (data >> corner_bits) & (0xFFFFFFFFu >> corner_bits);
      uint32_t corner = pair ? indices[2u * (size_t)(i + k) + 1u] : data & corner_mask;
      const orc_meshlet_instance* mli = &meshlet_instances[mli_index];
      const orc_mesh_instance* inst = &mesh_instances[mli->mesh_instance_index];
      const orc_mesh* mesh = &meshes[inst->mesh_index];
      const orc_mesh_lod* lod = &((const orc_mesh_lod*)(uintptr_t)mesh->lods)[inst->lod_index];
      const orc_meshlet* ml = &((const orc_meshlet*)(uintptr_t)lod->meshlets)[mli->meshlet_index];
      uint32_t li = micro_index((const uint32_t*)(uintptr_t)lod->local_triangle_indices, ml->local_triangle_index_offset + corner);
      uint32_t vi = ((const uint32_t*)(uintptr_t)lod->indirect_vertex_indices)[ml->indirect_vertex_index_offset + li];
      const uint16_t* pos = (const uint16_t*)(uintptr_t)mesh->vertex_positions;
      float p[3] = {orc_dequantize_half(pos[(size_t)vi * 4 + 0]), orc_dequantize_half(pos[(size_t)vi * 4 + 1]), orc_dequantize_half(pos[(size_t)vi * 4 + 2])};
      float world[4];
      mul_mp(xform(transforms, inst->transform_index), p, world);
      mul_mp(pv, world, poly[0][k]);
      if (k == 0) vis_out = (mli_index << 8) | ((corner / 3u) & 0xFFu); /* VisBufferData(mli, triangle_index / 3).encode() */
    }
    /* Sutherland-Hodgman against the five planes (the fixed-function clipper the graphics pipeline runs between vs_main and the
     * rasteriser; round 1 dropped every triangle with a corner at w <= 0).  A new vertex is always interpolated from the inside end
     * I of the crossing edge to its outside end O: t = d(I) / (d(I) - d(O)), v = I + t (O - I), so both triangles sharing the edge
     * compute the same vertex.  A triangle inside every plane passes through untouched. */
    int n = 3, cur = 0, crossed = 0;
    for (int pl = 0; pl < 5 && n >= 3; pl++) {
      int m = 0;
      for (int k = 0; k < n; k++) {
        const float* p = poly[cur][k];
        const float* q = poly[cur][(k + 1) % n];
        float dp = clip_distance(p, pl), dq = clip_distance(q, pl);
        int ip = dp >= 0.0f, iq = dq >= 0.0f;
        if (ip) {
          for (int c = 0; c < 4; c++) poly[cur ^ 1][m][c] = p[c];
          m++;
        }
        if (ip != iq) {
          crossed = 1;
          const float* I = ip ? p : q;
          const float* O = ip ? q : p;
          float dI = ip ? dp : dq, dO = ip ? dq : dp;
          float t = dI / (dI - dO);
          for (int c = 0; c < 4; c++) poly[cur ^ 1][m][c] = I[c] + t * (O[c] - I[c]);
          m++;
        }
      }
      n = m;
      cur ^= 1;
    }
    g_draw_clipped += (uint32_t)(crossed && n >= 3);
    for (int k = 1; k + 1 < n; k++) raster_clip_triangle(poly[cur][0], poly[cur][k], poly[cur][k + 1], vis_out, W, H, visdepth);
  }
}

void orc_resolve_visbuffer(const uint64_t* visdepth, uint32_t W, uint32_t H, float* depth, uint32_t* vis) {
  for (size_t i = 0; i < (size_t)W * H; i++) {
    depth[i] = u2f((uint32_t)(visdepth[i] >> 32));
    vis[i] = (uint32_t)visdepth[i];
  }
}
