// oxcull_device.hpp -- device-side arithmetic of the cull path, written for gfx950.
//
// This restates the reference shaders' arithmetic (Oxylus/src/Render/Shaders/cull.slang,
// scene.slang, common/math.slang) with the canonical IEEE-754 binary32 evaluation order of
// no FMA contraction (the TU is compiled with -ffp-contract=off and
// carries the pragma below), left-to-right dot / mat*vec, correctly rounded sqrt and divide
// (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt).  It is written independently of
// oracle/ (which is test infrastructure) -- the two must agree bit for bit, and the GPU parity
// tests check exactly that.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#pragma clang fp contract(off)

namespace oxc {

#define OXC_DEV __device__ __forceinline__

OXC_DEV float asf(uint32_t u) { return __builtin_bit_cast(float, u); }
OXC_DEV uint32_t asu(float f) { return __builtin_bit_cast(uint32_t, f); }
OXC_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
OXC_DEV uint32_t readlane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
OXC_DEV float readlane_f(uint32_t v, int l) { return asf(readlane_u(v, l)); }
OXC_DEV uint32_t readfirst_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
OXC_DEV float bperm_f(int src_lane, float v) {
  return asf((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)asu(v)));
}

// 64-bit device addresses coming out of the reference structs are integers; loading through a
// generic pointer would emit flat_load (LDS-aperture check, lgkmcnt).  Go through address space 1.
// Generic -> global pointer (pointers that come out of memory-resident argument blocks are generic
// to the compiler and would be accessed with flat_* instructions).
template <typename T>
OXC_DEV T __attribute__((address_space(1))) * gptr(T* p) {
  return (T __attribute__((address_space(1)))*)p;
}
template <typename T>
OXC_DEV const T __attribute__((address_space(1))) * gptr(const T* p) {
  return (const T __attribute__((address_space(1)))*)p;
}
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
OXC_DEV uint32_t load_global_u32(uint64_t base, uint32_t index) {
  return reinterpret_cast<const uint32_t __attribute__((address_space(1)))*>(base)[index];
}
OXC_DEV uint2 load_global_u2(uint64_t base, uint32_t index) {
  u32x2_t v = reinterpret_cast<const u32x2_t __attribute__((address_space(1)))*>(base)[index];
  return make_uint2(v.x, v.y);
}
OXC_DEV uint4 load_global_u4(uint64_t base, uint32_t index) {
  u32x4_t v = reinterpret_cast<const u32x4_t __attribute__((address_space(1)))*>(base)[index];
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Streamed-once variants (`nt`: the line is not kept in the caches; tools/bw_probe.hip measures 7.0 vs 6.2 TB/s for
// pure streaming reads on this chip).
OXC_DEV uint32_t load_stream_u32(uint64_t base, uint32_t index) {
  return __builtin_nontemporal_load(reinterpret_cast<const uint32_t __attribute__((address_space(1)))*>(base) + index);
}
OXC_DEV uint2 load_stream_u2(uint64_t base, uint32_t index) {
  u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t __attribute__((address_space(1)))*>(base) + index);
  return make_uint2(v.x, v.y);
}
OXC_DEV uint4 load_stream_u4(uint64_t base, uint32_t index) {
  u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t __attribute__((address_space(1)))*>(base) + index);
  return make_uint4(v.x, v.y, v.z, v.w);
}

OXC_DEV float dot3(float ax, float ay, float az, float bx, float by, float bz) { return (ax * bx + ay * by) + az * bz; }
OXC_DEV float len3(float x, float y, float z) { return __builtin_sqrtf(dot3(x, y, z, x, y, z)); }

// com::dequantize_half, common/math.slang:193-201 (h in the low 16 bits): denormals flush to signed
// zero, everything else is the IEEE value.  One instruction: v_cvt_f32_f16 with the wave's FP16/FP64
// denormal mode set to flush (set_half_denorm_flush() below, MODE.FP_DENORM[3:2] = 0) -- the hardware
// then reads a denormal half as +-0, which is exactly the shader's rule.  f32 denormal handling
// (MODE.FP_DENORM[1:0]) is untouched.  Only signalling-NaN inputs differ from the shader's bit
// formula (the conversion quiets them); no decision can depend on a NaN payload.
// tests/test_gpu_parity.py::test_decode_known_answers_all_halfs_and_s8 sweeps all 65536 inputs.
OXC_DEV float dequantize_half(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
// hwreg(HW_REG_MODE = 1, offset 6, width 2) = ((2 - 1) << 11) | (6 << 6) | 1
OXC_DEV void set_half_denorm_flush() { __builtin_amdgcn_s_setreg(((2 - 1) << 11) | (6 << 6) | 1, 0); }

// i32(s8) / 127.0 (scene.slang:408-418), correctly rounded.  x * (1/127) alone is wrong for 16
// of the 256 inputs; one Markstein refinement step is exact for all 256 (checked exhaustively
// on the CPU and by tests/test_gpu_kat.py on the device) and costs 3 VALU ops instead of a
// ~14-op IEEE division.
OXC_DEV float s8_over_127(int32_t x) {
  const float r = 1.0f / 127.0f;
  float xf = (float)x;
  float q = xf * r;
  float e = __builtin_fmaf(-q, 127.0f, xf);
  return __builtin_fmaf(e, r, q);
}

// Two binary32 values side by side: arithmetic on f2 is one packed VALU instruction (v_pk_mul_f32,
// v_pk_add_f32, v_pk_fma_f32) performing two independent IEEE operations -- the same roundings as the
// scalar spelling, half the issue slots.  A scalar broadcast {x, x} costs nothing (op_sel).
typedef float f2 __attribute__((ext_vector_type(2)));
OXC_DEV f2 splat(float x) { return f2{x, x}; }
// s8_over_127 for two values at once
OXC_DEV f2 s8_over_127_x2(int32_t x, int32_t y) {
  const f2 r = splat(1.0f / 127.0f);
  const f2 xf = {(float)x, (float)y};
  const f2 q = xf * r;
  const f2 e = __builtin_elementwise_fma(-q, splat(127.0f), xf);
  return __builtin_elementwise_fma(e, r, q);
}

// float -> u32 / i32, saturating, NaN -> 0: exactly what v_cvt_u32_f32 / v_cvt_i32_f32 do (truncation toward zero inside the
// range) -- tools/cvt_check.hip sweeps 2^27 bit patterns on the device against the spelled-out rule.  A C++ cast would be undefined out of
// range, and spelling the rule out compiled into two nested exec-mask branches per conversion: ~50 of the ~340 instructions of an
// occlusion batch in the HiZ meshlet kernels, which are bound by instruction issue.  Hence one instruction through an asm statement
// (VGPR in, VGPR out: no scalar hazards to pad).  (v_cvt_flr_i32_f32 does NOT equal cvt_i32(floor(x)) on this part: same sweep.)
OXC_DEV uint32_t cvt_u32_sat(float f) {
  uint32_t r;
  asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
OXC_DEV int32_t cvt_i32_sat(float f) {
  int32_t r;
  asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}

// Column-major 4x4 helpers: element (r,c) = m[c*4+r].
#define OXC_M(m, r, c) ((m)[(c)*4 + (r)])

OXC_DEV void mul_mat4(const float* a, const float* b, float* out) {
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      OXC_M(out, r, c) = ((OXC_M(a, r, 0) * OXC_M(b, 0, c) + OXC_M(a, r, 1) * OXC_M(b, 1, c)) + OXC_M(a, r, 2) * OXC_M(b, 2, c)) +
                         OXC_M(a, r, 3) * OXC_M(b, 3, c);
}

// cull.slang:49-71: the six normalised planes of mvp, in the shader's order.
OXC_DEV void frustum_planes(const float* mvp, float* pl /*24*/) {
  float r0[4], r1[4], r2[4], r3[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    r0[c] = OXC_M(mvp, 0, c);
    r1[c] = OXC_M(mvp, 1, c);
    r2[c] = OXC_M(mvp, 2, c);
    r3[c] = OXC_M(mvp, 3, c);
  }
  float t[6][4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    t[0][c] = r3[c] + r0[c];
    t[1][c] = r3[c] - r0[c];
    t[2][c] = r3[c] + r1[c];
    t[3][c] = r3[c] - r1[c];
    t[4][c] = r2[c];
    t[5][c] = r3[c] - r2[c];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float l = len3(t[i][0], t[i][1], t[i][2]);
#pragma unroll
    for (int c = 0; c < 4; c++) pl[i * 4 + c] = t[i][c] / l;
  }
}

// cull.slang:73-83 with pre-normalised planes, two planes per packed instruction.
// pl2[p*8 + c*2 + k] = component c of plane 2p+k; sg2[p*6 + c*2 + k] = +-1.0 carrying that component's
// sign bit.  The p-vertex component  c + asfloat(asuint(h) ^ signbit(n))  is  c + h * (+-1.0): the
// product is exact (also for -0, denormals, Inf; a NaN stays a NaN and fails every comparison either
// way), so the sum rounds once, exactly like the shader's.  Per plane pair: 3 pk_mul + 3 pk_add for the
// p-vertex, 3 pk_mul + 2 pk_add for the dot (left to right, no contraction), 2 compares.
OXC_DEV bool test_frustum_planes(const float* pl2, const float* sg2, float cx, float cy, float cz, float ex, float ey, float ez) {
  const f2 hxy = f2{ex, ey} * splat(0.5f);
  const float hz = ez * 0.5f;
  bool inside = true;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const float* n = pl2 + p * 8;
    const float* sg = sg2 + p * 6;
    const f2 qx = splat(cx) + splat(hxy.x) * f2{sg[0], sg[1]};
    const f2 qy = splat(cy) + splat(hxy.y) * f2{sg[2], sg[3]};
    const f2 qz = splat(cz) + splat(hz) * f2{sg[4], sg[5]};
    const f2 d = (qx * f2{n[0], n[1]} + qy * f2{n[2], n[3]}) + qz * f2{n[4], n[5]};
    inside = inside & !(d.x <= -n[6]) & !(d.y <= -n[7]);
  }
  return inside;
}

// The three packed plane distances test_frustum_planes compares (d[p] = {plane 2p, plane 2p + 1}), without the comparison: views whose
// planes have the SAME normals and signs -- the clipmap views of one light: row 3 of an orthographic matrix is (0, 0, 0, 1), the normalised
// normals are those of rows 0..2 and do not change with the clipmap's power-of-two scale -- differ only in the thresholds -n[6], -n[7], so
// the distances are computed once per box and compared once per view (k_cull_meshlets_hpb_test).  Same expressions, same roundings.
OXC_DEV void frustum_plane_dots(const float* pl2, const float* sg2, float cx, float cy, float cz, float ex, float ey, float ez, f2* d /*3*/) {
  const f2 hxy = f2{ex, ey} * splat(0.5f);
  const float hz = ez * 0.5f;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const float* n = pl2 + p * 8;
    const float* sg = sg2 + p * 6;
    const f2 qx = splat(cx) + splat(hxy.x) * f2{sg[0], sg[1]};
    const f2 qy = splat(cy) + splat(hxy.y) * f2{sg[2], sg[3]};
    const f2 qz = splat(cz) + splat(hz) * f2{sg[4], sg[5]};
    d[p] = (qx * f2{n[0], n[1]} + qy * f2{n[2], n[3]}) + qz * f2{n[4], n[5]};
  }
}

// One plane pair of test_frustum_planes (n8 = planes2[p], sg6 = signs2[p]): the same expressions, for callers that walk the pairs in a loop.
OXC_DEV bool frustum_pair_inside(const float* n, const float* sg, float cx, float cy, float cz, float ex, float ey, float ez) {
  const f2 hxy = f2{ex, ey} * splat(0.5f);
  const float hz = ez * 0.5f;
  const f2 qx = splat(cx) + splat(hxy.x) * f2{sg[0], sg[1]};
  const f2 qy = splat(cy) + splat(hxy.y) * f2{sg[2], sg[3]};
  const f2 qz = splat(cz) + splat(hz) * f2{sg[4], sg[5]};
  const f2 d = (qx * f2{n[0], n[1]} + qy * f2{n[2], n[3]}) + qz * f2{n[4], n[5]};
  return !(d.x <= -n[6]) & !(d.y <= -n[7]);
}

// scene.slang:292-299: cofactor matrix of world's upper 3x3, column-major 3x3 out.
OXC_DEV void normal_matrix(const float* w, float* nm) {
  float b[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < 3; k++) b[j][k] = OXC_M(w, k, j);
  const int a1[3] = {1, 2, 0}, a2[3] = {2, 0, 1};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float* u = b[a1[k]];
    const float* v = b[a2[k]];
    nm[k * 3 + 0] = u[1] * v[2] - v[1] * u[2];
    nm[k * 3 + 1] = u[2] * v[0] - v[2] * u[0];
    nm[k * 3 + 2] = u[0] * v[1] - v[0] * u[1];
  }
}

// Wave-uniform cone operands of one mesh instance (InstCache dwords 60, 64..85).
struct ConeU {
  float nm[9];       // normal matrix, column-major
  float w2[3][2];    // w2[c][k] = world(row k, col c), k = 0,1
  float wt2[2];      // world(row 0..1, col 3)
  float wr2[4];      // world row 2
  float scale_max;
};

// cull_meshlets.slang:49-52 + cull.slang:173-175.  Returns cone_visible.  Canonical (tier 2) arithmetic.
OXC_DEV bool cone_visible(const ConeU& u, float camx, float camy, float camz, float cx, float cy, float cz, float ex, float ey, float ez,
                          float ax, float ay, float az, float cutoff) {
  const float* nm = u.nm;
  float nx = (nm[0] * ax + nm[3] * ay) + nm[6] * az;
  float ny = (nm[1] * ax + nm[4] * ay) + nm[7] * az;
  float nz = (nm[2] * ax + nm[5] * ay) + nm[8] * az;
  float l = len3(nx, ny, nz);
  float kx = nx / l, ky = ny / l, kz = nz / l;
  float wx = ((u.w2[0][0] * cx + u.w2[1][0] * cy) + u.w2[2][0] * cz) + u.wt2[0];
  float wy = ((u.w2[0][1] * cx + u.w2[1][1] * cy) + u.w2[2][1] * cz) + u.wt2[1];
  float wz = ((u.wr2[0] * cx + u.wr2[1] * cy) + u.wr2[2] * cz) + u.wr2[3];
  float radius = len3(ex * 0.5f, ey * 0.5f, ez * 0.5f) * u.scale_max;
  float dx = wx - camx, dy = wy - camy, dz = wz - camz;
  bool culled = dot3(dx, dy, dz, kx, ky, kz) >= cutoff * len3(dx, dy, dz) + radius;
  return cutoff >= 1.0f || !culled;
}

// Two-tier cone test (cull_meshlets.slang:49-52, cull.slang:173-175).  Tier 1 evaluates
//   L = dot(d, n) / |n|   and   R = cutoff * |d| + |h| * scale
// with the 1-ulp hardware v_rsq_f32 / v_sqrt_f32 (quarter-rate single instructions) instead of three
// IEEE divisions and three correctly rounded square roots.  n, d and h are formed with the canonical
// roundings (x/y components as packed pairs -- same operations); from there both tiers approximate the
// same real numbers with a relative error of a few 2^-23 of (|d| + radius) -- at most ~12 roundings of
// magnitude <= |d| + radius on either side, i.e. < 3e-6 * (|d| + radius) -- so when |L - R| exceeds
// kConeMargin * (|d| + radius) = 1.6e-5 * (...) the canonical (tier 2) comparison is already decided.
// Lanes inside the margin are undecided; the caller runs the exact path when any lane of the wave is.
// Returns: 0 = culled, 1 = cone-visible, 2 = undecided.
constexpr float kConeMargin = 1.6e-5f;
OXC_DEV int cone_visible_fast(const ConeU& u, float camx, float camy, float camz, float cx, float cy, float cz, float ex, float ey, float ez,
                              float ax, float ay, float az, float cutoff) {
  const float* nm = u.nm;
  const f2 nxy = (f2{nm[0], nm[1]} * splat(ax) + f2{nm[3], nm[4]} * splat(ay)) + f2{nm[6], nm[7]} * splat(az);
  const float nz = (nm[2] * ax + nm[5] * ay) + nm[8] * az;
  const f2 wxy = ((f2{u.w2[0][0], u.w2[0][1]} * splat(cx) + f2{u.w2[1][0], u.w2[1][1]} * splat(cy)) + f2{u.w2[2][0], u.w2[2][1]} * splat(cz)) +
                 f2{u.wt2[0], u.wt2[1]};
  const float wz = ((u.wr2[0] * cx + u.wr2[1] * cy) + u.wr2[2] * cz) + u.wr2[3];
  const f2 dxy = wxy - f2{camx, camy};
  const float dz = wz - camz;
  const f2 hxy = f2{ex, ey} * splat(0.5f);
  const float hz = ez * 0.5f;
  const f2 nn = nxy * nxy, dd = dxy * dxy, dn = dxy * nxy, hh = hxy * hxy;
  float inv_l = __builtin_amdgcn_rsqf((nn.x + nn.y) + nz * nz);
  float dlen = __builtin_amdgcn_sqrtf((dd.x + dd.y) + dz * dz);
  float radius = __builtin_amdgcn_sqrtf((hh.x + hh.y) + hz * hz) * u.scale_max;
  float L = ((dn.x + dn.y) + dz * nz) * inv_l;
  float R = cutoff * dlen + radius;
  float T = kConeMargin * (dlen + radius);
  float diff = L - R;
  // NaN/Inf anywhere (degenerate axis, huge values) compares false twice -> undecided -> exact path
  if (diff > T) return 0;
  if (diff < -T) return 1;
  return 2;
}

struct HizView {
  const float* data;
  uint32_t width, height, levels;
  // Top of the pyramid staged in LDS: levels >= lds_first live at lds + lds_off[level]
  // (lds_first == levels: nothing staged).  Same values, different address space.
  const float* lds;
  const uint32_t* lds_off;
  uint32_t lds_first;
  // x / width == x * (1 / width) bit for bit when width is a power of two (the reference sizes the pyramid with bit_ceil,
  // RendererInstance.cpp:573-577): one multiply instead of a ~12-instruction IEEE division per axis.  0: not a power of two -> divide.
  float inv_width, inv_height;
};
// 1 / d when d is a power of two (exact), else 0
OXC_DEV float exact_reciprocal_or_zero(uint32_t d) { return (d != 0u && (d & (d - 1u)) == 0u) ? 1.0f / (float)d : 0.0f; }

OXC_DEV uint32_t mip_dim(uint32_t d, uint32_t mip) {
  uint32_t v = d >> mip;
  return v ? v : 1u;
}

// cull.slang:12-47 project_aabb.  Returns false for `none` (box crosses the near plane).
// out = {min.u, min.v, min.z, max.u, max.v, max.z}.
// TRY_AFFINE (callers whose matrices are usually orthographic -- the clipmap views of a directional light): when the matrix's last
// row is exactly (0, 0, 0, 1) every finite corner has w == 1.0f exactly (0 * x is +-0, 1 + +-0 is 1), a division by 1 returns its
// numerator, and the fold over the eight corners collapses: corner (bx, by, bz) is fl(fl(fl(P0 + bx SX) + by SY) + bz SZ), rounding is
// monotone in each addend and adding 0 is exact, so min over the corners = ((P0 + min(0, SX)) + min(0, SY)) + min(0, SZ) and max
// likewise -- the same floats as the 24 divisions and 48 min / max of the general path (up to the sign of a zero, which neither
// q * 0.5 + 0.5 nor the depth comparison sees) for 36 instructions.  Lanes whose sums could leave the finite range (any |P0| + |SX| +
// |SY| + |SZ| above 2^120, or a NaN) send the wave down the general path.  (tests: test_project_aabb_matches_ieee_division_bit_for_bit)
// NEED_Z = false (test_vsm_page reads only the four uv bounds, cull.slang:137-166): the orthographic branch leaves the z row out -- neither
// the uv bounds nor the `none` decision depend on it there (every w is 1; a z sum that leaves the finite range changes out[2] / out[5] only,
// and a NaN / Inf INPUT coordinate reaches the x and y rows as well, 0 * Inf being NaN, so their finiteness test still sends the wave down
// the general path).  out[2] / out[5] are then unspecified.  Round 6: 13 of the kernel's ~215 VALU instructions per (batch, view).
// v_min / v_max against 0 as single instructions (fminf / fmaxf put a canonicalising v_max_f32 x, x in front of each: the operands here are
// finite by the test above them).
OXC_DEV float min0_finite(float x) {
  float r;
  asm("v_min_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
OXC_DEV float max0_finite(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
template <bool TRY_AFFINE = false, bool NEED_Z = true>
OXC_DEV bool project_aabb(const float* mvp, float near_clip, float cx, float cy, float cz, float ex, float ey, float ez, float* out) {
  float SX[4], SY[4], SZ[4], P[8][4];
  float p0x = cx - ex * 0.5f, p0y = cy - ey * 0.5f, p0z = cz - ez * 0.5f;
  if (TRY_AFFINE && asu(OXC_M(mvp, 3, 0)) == 0u && asu(OXC_M(mvp, 3, 1)) == 0u && asu(OXC_M(mvp, 3, 2)) == 0u &&
      asu(OXC_M(mvp, 3, 3)) == 0x3F800000u) {  // (the matrix is wave-uniform in every caller that sets TRY_AFFINE)
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f}, P0[3] = {0.f, 0.f, 0.f}, S[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    bool finite = true;
#pragma unroll
    for (int i = 0; i < (NEED_Z ? 3 : 2); i++) {
      S[i][0] = OXC_M(mvp, i, 0) * ex, S[i][1] = OXC_M(mvp, i, 1) * ey, S[i][2] = OXC_M(mvp, i, 2) * ez;
      P0[i] = ((OXC_M(mvp, i, 0) * p0x + OXC_M(mvp, i, 1) * p0y) + OXC_M(mvp, i, 2) * p0z) + OXC_M(mvp, i, 3);
      finite = finite && ((__builtin_fabsf(P0[i]) + __builtin_fabsf(S[i][0])) + (__builtin_fabsf(S[i][1]) + __builtin_fabsf(S[i][2]))) <= 1.329227995784916e36f;
    }
    if (__builtin_amdgcn_ballot_w64(!finite) == 0) {
#pragma unroll
      for (int i = 0; i < (NEED_Z ? 3 : 2); i++) {  // (every operand is finite here)
        lo[i] = ((P0[i] + min0_finite(S[i][0])) + min0_finite(S[i][1])) + min0_finite(S[i][2]);
        hi[i] = ((P0[i] + max0_finite(S[i][0])) + max0_finite(S[i][1])) + max0_finite(S[i][2]);
      }
      if (1.0f < near_clip) return false;  // every w is 1
      out[0] = lo[0] * 0.5f + 0.5f;
      out[1] = lo[1] * 0.5f + 0.5f;
      out[2] = lo[2];
      out[3] = hi[0] * 0.5f + 0.5f;
      out[4] = hi[1] * 0.5f + 0.5f;
      out[5] = hi[2];
      return true;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    SX[i] = OXC_M(mvp, i, 0) * ex;
    SY[i] = OXC_M(mvp, i, 1) * ey;
    SZ[i] = OXC_M(mvp, i, 2) * ez;
    P[0][i] = ((OXC_M(mvp, i, 0) * p0x + OXC_M(mvp, i, 1) * p0y) + OXC_M(mvp, i, 2) * p0z) + OXC_M(mvp, i, 3);
  }
#pragma unroll
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
#pragma unroll
  for (int k = 6; k >= 0; k--) depth = fminf(P[k][3], depth);
  if (depth < near_clip) return false;
  float vmin[3], vmax[3];
  // ---- the 24 perspective divisions.  A correctly rounded f32 division is, on this hardware, v_div_scale x2,
  // v_rcp, two FMAs refining the reciprocal, one multiply and three FMAs refining the quotient, v_div_fmas,
  // v_div_fixup (~12 instructions).  The scale / fixup steps only act outside a wide exponent window; inside it
  // the sequence is the plain FMA chain below, and the reciprocal part depends on the denominator alone -- so the
  // three quotients of a corner share it, and x / y run as one packed chain: bit-identical quotients for ~5
  // instructions each.  The window is checked per lane (all |numerators| <= 2^60, every z numerator >= 2^-60 in
  // magnitude, all w in [2^-30, 2^60]); x / y numerators below the window only produce |q| < 2^-25, which the
  // caller's q * 0.5 + 0.5 maps to exactly 0.5 on either path.  If any lane of the wave is outside the window the
  // wave takes the IEEE divisions.  (tests: test_project_aabb_matches_ieee_division_bit_for_bit)
  float amax = 0.0f, zmin = 3.402823466e+38f, wmax = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    amax = fmaxf(amax, fmaxf(fmaxf(__builtin_fabsf(P[k][0]), __builtin_fabsf(P[k][1])), __builtin_fabsf(P[k][2])));
    zmin = fminf(zmin, __builtin_fabsf(P[k][2]));
    wmax = fmaxf(wmax, P[k][3]);
  }
  const bool in_window = amax <= 1.152921504606846976e18f && zmin >= 8.673617379884035e-19f && depth >= 9.313225746154785e-10f &&
                         wmax <= 1.152921504606846976e18f;
  if (__builtin_amdgcn_ballot_w64(!in_window) == 0) {
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 7; k >= 0; k--) {  // corner 7 first, then 6..0: the fold order of the IEEE branch below
      const float w = P[k][3];
      const float r0 = __builtin_amdgcn_rcpf(w);
      const float e = __builtin_fmaf(-w, r0, 1.0f);
      const float r = __builtin_fmaf(e, r0, r0);
      const f2 n = {P[k][0], P[k][1]};
      const f2 nw = splat(-w), rr = splat(r);
      f2 q = n * rr;
      f2 e1 = __builtin_elementwise_fma(nw, q, n);
      q = __builtin_elementwise_fma(e1, rr, q);
      e1 = __builtin_elementwise_fma(nw, q, n);
      q = __builtin_elementwise_fma(e1, rr, q);
      const float nz = P[k][2];
      float z = nz * r;
      float ez = __builtin_fmaf(-w, z, nz);
      z = __builtin_fmaf(ez, r, z);
      ez = __builtin_fmaf(-w, z, nz);
      z = __builtin_fmaf(ez, r, z);
      if (k == 7) {
        lo[0] = hi[0] = q.x;
        lo[1] = hi[1] = q.y;
        lo[2] = hi[2] = z;
      } else {
        lo[0] = fminf(q.x, lo[0]);
        hi[0] = fmaxf(q.x, hi[0]);
        lo[1] = fminf(q.y, lo[1]);
        hi[1] = fmaxf(q.y, hi[1]);
        lo[2] = fminf(z, lo[2]);
        hi[2] = fmaxf(z, hi[2]);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      vmin[j] = lo[j];
      vmax[j] = hi[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float lo = P[7][j] / P[7][3];
      float hi = lo;
#pragma unroll
      for (int k = 6; k >= 0; k--) {
        float d = P[k][j] / P[k][3];
        lo = fminf(d, lo);
        hi = fmaxf(d, hi);
      }
      vmin[j] = lo;
      vmax[j] = hi;
    }
  }
  out[0] = vmin[0] * 0.5f + 0.5f;
  out[1] = vmin[1] * 0.5f + 0.5f;
  out[2] = vmin[2];
  out[3] = vmax[0] * 0.5f + 0.5f;
  out[4] = vmax[1] * 0.5f + 0.5f;
  out[5] = vmax[2];
  return true;
}

// cull.slang:12-47 + :86-135.  Returns true when the box is occluded; a box for which
// project_aabb returns none stays visible (cull_meshlets_hiz.slang:61-65).  Inactive lanes
// return before touching memory.  level_off: float offsets of each mip.
OXC_DEV bool aabb_occluded(const float* mvp, float near_clip, float cx, float cy, float cz, float ex, float ey, float ez,
                           const HizView& hiz, const uint32_t* level_off, bool active) {
  float sa[6];
  if (!active) return false;
  if (!project_aabb(mvp, near_clip, cx, cy, cz, ex, ey, ez, sa)) return false;
  const float minu = sa[0], minv = sa[1], maxu = sa[3], maxv = sa[4], maxz = sa[5];

  // test_occlusion, cull.slang:114-135
  float sw = (float)hiz.width, sh = (float)hiz.height;
  uint32_t minx = cvt_u32_sat(fmaxf(minu * sw, 0.0f));
  uint32_t miny = cvt_u32_sat(fmaxf(minv * sh, 0.0f));
  uint32_t maxx = cvt_u32_sat(fminf(maxu * sw, sw - 1.0f));
  uint32_t maxy = cvt_u32_sat(fminf(maxv * sh, sh - 1.0f));
  uint32_t szx = maxx - minx, szy = maxy - miny;  // u32 wrap-around (cull.slang:127)
  uint32_t ms = szx > szy ? szx : szy;
  // ceil(log2(float(ms))) in integers, clamped to [0, levels-1]
  uint32_t mip = ms <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(ms - 1u);
  uint32_t top = hiz.levels - 1u;
  mip = mip > top ? top : mip;
  const float uc = ((float)minx + (float)maxx) * 0.5f, vc = ((float)miny + (float)maxy) * 0.5f;
  float u = hiz.inv_width != 0.0f ? uc * hiz.inv_width : uc / sw;  // (wave-uniform choice)
  float v = hiz.inv_height != 0.0f ? vc * hiz.inv_height : vc / sh;

  // sample_level_min_reduction_2x2, cull.slang:86-112
  uint32_t mw = mip_dim(hiz.width, mip), mh = mip_dim(hiz.height, mip);
  int32_t mxx = (int32_t)mw - 1, mxy = (int32_t)mh - 1;
  int32_t bx = cvt_i32_sat(floorf(u * (float)mw - 0.5f));
  int32_t by = cvt_i32_sat(floorf(v * (float)mh - 0.5f));
  int32_t bx1 = (int32_t)((uint32_t)bx + 1u), by1 = (int32_t)((uint32_t)by + 1u);
  int32_t x0 = min(max(bx, 0), mxx), y0 = min(max(by, 0), mxy);
  int32_t x1 = min(max(bx1, 0), mxx), y1 = min(max(by1, 0), mxy);
  float p00, p10, p01, p11;
  if (mip >= hiz.lds_first) {  // small, heavily shared top mips: LDS-staged tile of the pyramid
    const float* lvl = hiz.lds + hiz.lds_off[mip];
    p00 = lvl[y0 * (int32_t)mw + x0];
    p10 = lvl[y0 * (int32_t)mw + x1];
    p01 = lvl[y1 * (int32_t)mw + x0];
    p11 = lvl[y1 * (int32_t)mw + x1];
  } else {
    const float* lvl = hiz.data + level_off[mip];
    p00 = lvl[(size_t)y0 * mw + x0];
    p10 = lvl[(size_t)y0 * mw + x1];
    p01 = lvl[(size_t)y1 * mw + x0];
    p11 = lvl[(size_t)y1 * mw + x1];
  }
  float d = fminf(fminf(p00, p10), fminf(p01, p11));
  return maxz <= d - 1e-7f;
}

// Hierarchical page buffer: R8UI Texture2DArray with mips, linear (level k: `layers` planes of
// max(1,w>>k) x max(1,h>>k) bytes at data + level_off[k]).
struct HpbView {
  const uint8_t* data;
  uint32_t width, height, layers, levels;
};

// ceil(log2(x)) of a float from its bits (exact), clamped to [0, levels-1]; x <= 0 / NaN -> 0.
OXC_DEV uint32_t ceil_log2f_clamped(float x, uint32_t levels) {
  // straight-line (round 6: the two early returns were exec-masked branches in the VSM page test): a denormal's exponent field is 0, so its
  // c is negative and the lower clamp gives the 0 the early return gave; x <= 0 and NaN are the one select at the end
  const uint32_t b = asu(x);
  int32_t c = (int32_t)((b >> 23) & 0xFFu) - 127 + ((b & 0x7FFFFFu) ? 1 : 0);
  c = min(max(c, 0), (int32_t)levels - 1);
  return (x > 0.0f) ? (uint32_t)c : 0u;
}
OXC_DEV float fract_f(float x) { return x - floorf(x); }

// cull.slang:137-166 test_vsm_page (nearest, clamped SampleLevel at an integral mip).  The four taps share the level (its extent, its
// layer's first byte) and pairwise their column / row; a pyramid is at most 4096^2 x 16 layers with its mips = 358 MB, so byte offsets
// are 32-bit (oxc_cull_geometry checks that) and a tap's address is one 64-bit add (round 6: the 64-bit multiply-adds per tap were a
// fifth of the kernel's page-test instructions).
OXC_DEV bool test_vsm_page(const float* a, const HpbView& h, const uint32_t* level_off, uint32_t layer, int32_t pox_i, int32_t poy_i) {
  float sw = (float)h.width, sh = (float)h.height;
  float pox = (float)pox_i / sw, poy = (float)poy_i / sh;
  float box_w = (a[3] - a[0]) * sw, box_h = (a[4] - a[1]) * sh;
  uint32_t mip = ceil_log2f_clamped(fmaxf(box_w, box_h), h.levels);
  const uint32_t mw = mip_dim(h.width, mip), mh = mip_dim(h.height, mip);
  const float fw = (float)mw, fh = (float)mh;
  // clamp(i32(floor(t * extent)), 0, extent - 1) for t = x - floor(x): t is in [0, 1] or NaN (x = +-Inf / NaN), so the product is >= 0 or NaN -- the
  // conversion truncates (= floor for >= 0) and turns NaN into 0: neither the floor nor the lower clamp can act (round 6: 8 instructions per page test)
  auto texel = [](float t, float extent_f, uint32_t extent) { return (uint32_t)min(cvt_i32_sat(t * extent_f), (int32_t)extent - 1); };
  const uint32_t x0 = texel(fract_f(a[0] + pox), fw, mw), x1 = texel(fract_f(a[3] + pox), fw, mw);
  const uint32_t y0 = texel(fract_f(a[1] + poy), fh, mh), y1 = texel(fract_f(a[4] + poy), fh, mh);
  const uint32_t base = level_off[mip] + layer * mw * mh, r0 = base + y0 * mw, r1 = base + y1 * mw;
  const uint32_t tl = h.data[r0 + x0], tr = h.data[r0 + x1], bl = h.data[r1 + x0], br = h.data[r1 + x1];
  return (tl | tr | bl | br) != 0u;
}

// cull_meshlets_hpb.slang:53-54 + cull.slang:177-179: directional cone test.
OXC_DEV bool cone_visible_directional(const float* nm, float dirx, float diry, float dirz, float ax, float ay, float az, float cutoff) {
  float nx = (nm[0] * ax + nm[3] * ay) + nm[6] * az;
  float ny = (nm[1] * ax + nm[4] * ay) + nm[7] * az;
  float nz = (nm[2] * ax + nm[5] * ay) + nm[8] * az;
  float l = len3(nx, ny, nz);
  float kx = nx / l, ky = ny / l, kz = nz / l;
  bool culled = dot3(kx, ky, kz, dirx, diry, dirz) >= cutoff;
  return cutoff >= 1.0f || !culled;
}

}  // namespace oxc
