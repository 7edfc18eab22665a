// oxcull_meshbuild.cpp -- asset-side producer of the arrays the cull path consumes (the part round 1 left to a test
// stand-in): triangle soup -> LOD chain -> meshlets, host code as in the reference.
//
// Replaces the per-LOD loop of Oxylus/src/Asset/AssetManager_GLTF.cpp:599-682:
//     lod 0 = the input indices; lod i = simplify(lod i-1) to HALF its index count (((n + 5) / 6) * 3), border locked,
//     error accumulated; the chain stops when a LOD misses its target by more than 50 %, its error exceeds 0.5 or fewer than
//     two triangles remain (:639-645), or after GPU::Mesh::MAX_LODS = 8 levels;
//     every LOD is cut into meshlets of <= 64 vertices / <= 64 triangles (Model::MAX_MESHLET_INDICES / _PRIMITIVES, cone_weight 0)
//     whose u8 micro-index runs start 4-byte aligned (:687).
// The two algorithms themselves live in meshoptimizer v1.2 (xmake/packages.lua:9: meshopt_simplifyWithAttributes,
// meshopt_buildMeshlets), which is NOT under /root/reference and not in this image.  What is restated here is their published
// shape -- greedy edge collapse onto an endpoint ordered by a quadric error with an attribute (normal) term, error relative
// to the mesh extent; greedy meshlet growth over vertex adjacency preferring triangles that add no vertex -- not their exact
// heuristics, so the OUTPUT differs from meshoptimizer's triangle for triangle.  Nothing downstream depends on which valid
// clustering it gets: the contract that is tested is the format + validity (every triangle of a LOD in exactly one meshlet, limits,
// alignment, locked border, halving LODs, monotone errors, determinism) and that the result flows through the bounds producer,
// the mesh blob and cull_meshes / cull_meshlets / cull_triangles to the checker's bytes.
// (meshopt_optimizeVertexCache, :648-653, only reorders triangles for the post-transform cache of a hardware rasteriser; the
// meshlet builder below walks adjacency, not input order, so it is not restated.)
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <queue>
#include <unordered_map>
#include <vector>

#include "oxcull.h"
#include "oxcull_types.hpp"

namespace {
using oxc::GpuMeshlet;

struct V3 {
  double x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Symmetric 4x4 quadric {a2 ab ac ad b2 bc bd c2 cd d2} + its weight (sum of triangle areas)
struct Quadric {
  double q[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double w = 0;
  void add_plane(double a, double b, double c, double d, double weight) {
    const double v[10] = {a * a, a * b, a * c, a * d, b * b, b * c, b * d, c * c, c * d, d * d};
    for (int i = 0; i < 10; i++) q[i] += v[i] * weight;
    w += weight;
  }
  void add(const Quadric& o) {
    for (int i = 0; i < 10; i++) q[i] += o.q[i];
    w += o.w;
  }
  double eval(V3 p) const {
    return q[0] * p.x * p.x + 2 * q[1] * p.x * p.y + 2 * q[2] * p.x * p.z + 2 * q[3] * p.x + q[4] * p.y * p.y + 2 * q[5] * p.y * p.z + 2 * q[6] * p.y +
           q[7] * p.z * p.z + 2 * q[8] * p.z + q[9];
  }
};

struct Simplifier {
  const float* pos;
  const float* nrm;  // may be null
  uint32_t vcount;
  double extent = 1.0;
  std::vector<std::array<uint32_t, 3>> tris;
  std::vector<uint8_t> alive;
  std::vector<std::vector<uint32_t>> inc;  // vertex -> incident triangles (may hold dead ones: filtered on use)
  std::vector<Quadric> quad;
  std::vector<uint8_t> locked;
  std::vector<uint32_t> version;
  uint32_t live = 0;

  V3 P(uint32_t v) const { return {pos[v * 3], pos[v * 3 + 1], pos[v * 3 + 2]}; }

  void init(const uint32_t* idx, size_t n) {
    tris.clear();
    for (size_t i = 0; i + 2 < n; i += 3)
      if (idx[i] != idx[i + 1] && idx[i + 1] != idx[i + 2] && idx[i] != idx[i + 2]) tris.push_back({idx[i], idx[i + 1], idx[i + 2]});
    live = (uint32_t)tris.size();
    alive.assign(tris.size(), 1);
    inc.assign(vcount, {});
    quad.assign(vcount, Quadric());
    locked.assign(vcount, 0);
    version.assign(vcount, 0);
    V3 lo = {1e300, 1e300, 1e300}, hi = {-1e300, -1e300, -1e300};
    std::unordered_map<uint64_t, uint32_t> edge_use;
    edge_use.reserve(tris.size() * 3);
    for (uint32_t t = 0; t < tris.size(); t++) {
      const auto& tr = tris[t];
      const V3 a = P(tr[0]), b = P(tr[1]), c = P(tr[2]);
      V3 n = cross(b - a, c - a);
      const double len = std::sqrt(dot(n, n));
      if (len > 0) {
        n = {n.x / len, n.y / len, n.z / len};
        const double d = -dot(n, a), area = 0.5 * len;
        for (int k = 0; k < 3; k++) quad[tr[k]].add_plane(n.x, n.y, n.z, d, area);
      }
      for (int k = 0; k < 3; k++) {
        inc[tr[k]].push_back(t);
        const V3 p = P(tr[k]);
        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
        const uint32_t u = tr[k], v = tr[(k + 1) % 3];
        edge_use[((uint64_t)std::min(u, v) << 32) | std::max(u, v)]++;
      }
    }
    extent = tris.empty() ? 1.0 : std::max({hi.x - lo.x, hi.y - lo.y, hi.z - lo.z, 1e-30});
    for (auto& kv : edge_use)  // meshopt_SimplifyLockBorder: vertices on an open edge do not move
      if (kv.second == 1) locked[(uint32_t)(kv.first >> 32)] = locked[(uint32_t)kv.first] = 1;
  }

  // error of moving vertex i onto vertex j, relative to the extent; < 0: not allowed (flips a triangle)
  double cost(uint32_t i, uint32_t j) const {
    const V3 pj = P(j), pi = P(i);
    {
      // link condition: the vertices adjacent to BOTH ends must be exactly the apexes of the triangles on the edge, or the
      // collapse pinches the surface (an interior edge ends up with one or three triangles: a new border the lock cannot see)
      uint32_t ni[64], nj[64], apex[8];
      int ci = 0, cj = 0, ca = 0;
      auto gather = [&](uint32_t v, uint32_t other, uint32_t* out, int& n) {
        for (uint32_t t : inc[v]) {
          if (!alive[t]) continue;
          const auto& tr = tris[t];
          const bool on_edge = tr[0] == other || tr[1] == other || tr[2] == other;
          for (int k = 0; k < 3; k++) {
            const uint32_t u = tr[k];
            if (u == v || u == other) continue;
            if (on_edge && v == i) {
              bool seen = false;
              for (int a = 0; a < ca; a++) seen |= apex[a] == u;
              if (!seen) {
                if (ca == 8) return false;
                apex[ca++] = u;
              }
            }
            bool seen = false;
            for (int a = 0; a < n; a++) seen |= out[a] == u;
            if (!seen) {
              if (n == 64) return false;  // a vertex of very high valence: leave it alone
              out[n++] = u;
            }
          }
        }
        return true;
      };
      if (!gather(i, j, ni, ci) || !gather(j, i, nj, cj)) return -1.0;
      int common = 0;
      for (int a = 0; a < ci; a++)
        for (int b = 0; b < cj; b++)
          if (ni[a] == nj[b]) {
            common++;
            bool is_apex = false;
            for (int c = 0; c < ca; c++) is_apex |= apex[c] == ni[a];
            if (!is_apex) return -1.0;
          }
      if (common != ca || ca == 0 || ca > 2) return -1.0;
    }
    for (uint32_t t : inc[i]) {
      if (!alive[t]) continue;
      const auto& tr = tris[t];
      if (tr[0] == j || tr[1] == j || tr[2] == j) continue;  // collapses away
      V3 c[3], m[3];
      for (int k = 0; k < 3; k++) {
        c[k] = P(tr[k]);
        m[k] = tr[k] == i ? pj : c[k];
      }
      const V3 n0 = cross(c[1] - c[0], c[2] - c[0]), n1 = cross(m[1] - m[0], m[2] - m[0]);
      if (dot(n0, n1) <= 0.25 * std::sqrt(dot(n0, n0) * dot(n1, n1))) return -1.0;  // turns by more than ~75 degrees or degenerates
    }
    double e = std::max(quad[i].eval(pj), 0.0);
    if (nrm) {  // attribute term: a full normal reversal weighs like a displacement of 2 % of the extent over the vertex' area
      const double dx = nrm[i * 3] - nrm[j * 3], dy = nrm[i * 3 + 1] - nrm[j * 3 + 1], dz = nrm[i * 3 + 2] - nrm[j * 3 + 2];
      e += quad[i].w * (dx * dx + dy * dy + dz * dz) * (0.01 * extent) * (0.01 * extent);
    }
    (void)pi;
    return std::sqrt(e / std::max(quad[i].w, 1e-300)) / extent;
  }

  struct Cand {
    double cost;
    uint32_t i, j, vi, vj;
    bool operator<(const Cand& o) const {  // min-heap on (cost, i, j): deterministic
      if (cost != o.cost) return cost > o.cost;
      if (i != o.i) return i > o.i;
      return j > o.j;
    }
  };

  void push_edges(std::priority_queue<Cand>& heap, uint32_t v) {
    for (uint32_t t : inc[v]) {
      if (!alive[t]) continue;
      for (int k = 0; k < 3; k++) {
        const uint32_t u = tris[t][k];
        if (u == v) continue;
        if (!locked[v]) {
          const double c = cost(v, u);
          if (c >= 0) heap.push({c, v, u, version[v], version[u]});
        }
        if (!locked[u]) {
          const double c = cost(u, v);
          if (c >= 0) heap.push({c, u, v, version[u], version[v]});
        }
      }
    }
  }

  // returns the worst relative error of the collapses performed
  double run(size_t target_index_count) {
    std::priority_queue<Cand> heap;
    for (uint32_t v = 0; v < vcount; v++)
      if (!inc[v].empty() && !locked[v])
        for (uint32_t t : inc[v])
          for (int k = 0; k < 3; k++) {
            const uint32_t u = tris[t][k];
            if (u == v) continue;
            const double c = cost(v, u);
            if (c >= 0) heap.push({c, v, u, 0, 0});
          }
    double worst = 0;
    while (!heap.empty() && (size_t)live * 3 > target_index_count) {
      const Cand c = heap.top();
      heap.pop();
      if (c.vi != version[c.i] || c.vj != version[c.j] || locked[c.i]) continue;
      const double now = cost(c.i, c.j);
      if (now < 0) continue;
      if (now > c.cost * (1 + 1e-9) + 1e-300) {  // stale estimate: requeue at its current price
        heap.push({now, c.i, c.j, c.vi, c.vj});
        continue;
      }
      worst = std::max(worst, now);
      for (uint32_t t : inc[c.i]) {
        if (!alive[t]) continue;
        auto& tr = tris[t];
        if (tr[0] == c.j || tr[1] == c.j || tr[2] == c.j) {
          alive[t] = 0;
          live--;
          continue;
        }
        for (int k = 0; k < 3; k++)
          if (tr[k] == c.i) tr[k] = c.j;
        inc[c.j].push_back(t);
      }
      inc[c.i].clear();
      quad[c.j].add(quad[c.i]);
      version[c.i]++;
      version[c.j]++;
      // compact j's list now and then so it does not grow without bound
      auto& lj = inc[c.j];
      lj.erase(std::remove_if(lj.begin(), lj.end(), [&](uint32_t t) { return !alive[t]; }), lj.end());
      std::sort(lj.begin(), lj.end());
      lj.erase(std::unique(lj.begin(), lj.end()), lj.end());
      // Re-price the edges at j (both directions).  Other edges of j's neighbours keep their queued price: every candidate is
      // priced again when it is popped and goes back into the queue if it got more expensive, so a stale entry can only be tried
      // late, never wrongly (re-pricing whole neighbourhoods here made a 159 K-triangle mesh take 12 s).
      push_edges(heap, c.j);
    }
    return worst;
  }

  std::vector<uint32_t> indices() const {
    std::vector<uint32_t> out;
    out.reserve((size_t)live * 3);
    for (uint32_t t = 0; t < tris.size(); t++)
      if (alive[t]) out.insert(out.end(), tris[t].begin(), tris[t].end());
    return out;
  }
};

struct Lod {
  std::vector<uint32_t> indices;
  std::vector<GpuMeshlet> meshlets;
  std::vector<uint32_t> ivi;
  std::vector<uint8_t> lti;
  float error = 0.f;
};

// Greedy meshlet growth over vertex adjacency (the shape of meshopt_buildMeshlets with cone_weight 0): among the unused triangles
// that touch the current meshlet take the one needing the fewest new vertices, then the one whose corners have the fewest unused
// triangles left (it would otherwise be stranded), then the nearest to the meshlet's centroid; when none touches it, restart from
// the unused triangle nearest to the centroid.  A meshlet closes when the next triangle does not fit.
void build_meshlets(const std::vector<uint32_t>& idx, const float* pos, uint32_t vcount, uint32_t max_v, uint32_t max_t, Lod& out) {
  const uint32_t T = (uint32_t)(idx.size() / 3);
  std::vector<uint32_t> off(vcount + 1, 0), adj(idx.size());
  for (uint32_t i : idx) off[i + 1]++;
  for (uint32_t v = 0; v < vcount; v++) off[v + 1] += off[v];
  {
    std::vector<uint32_t> fill(off.begin(), off.end() - 1);
    for (uint32_t t = 0; t < T; t++)
      for (int k = 0; k < 3; k++) adj[fill[idx[t * 3 + k]]++] = t;
  }
  std::vector<uint32_t> live(vcount, 0);
  for (uint32_t v = 0; v < vcount; v++) live[v] = off[v + 1] - off[v];
  std::vector<uint8_t> used(T, 0);
  std::vector<uint8_t> slot(vcount, 0xFF);
  std::vector<V3> cen(T);
  for (uint32_t t = 0; t < T; t++) {
    V3 c = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
      const uint32_t v = idx[t * 3 + k];
      c.x += pos[v * 3], c.y += pos[v * 3 + 1], c.z += pos[v * 3 + 2];
    }
    cen[t] = {c.x / 3, c.y / 3, c.z / 3};
  }
  // restart order: unused triangles by a coarse 3-D grid cell of their centroid (nearest-unused search walks outward from the
  // centroid's cell)
  V3 lo = {1e300, 1e300, 1e300}, hi = {-1e300, -1e300, -1e300};
  for (const V3& c : cen) {
    lo = {std::min(lo.x, c.x), std::min(lo.y, c.y), std::min(lo.z, c.z)};
    hi = {std::max(hi.x, c.x), std::max(hi.y, c.y), std::max(hi.z, c.z)};
  }
  const int G = (int)std::max(1.0, std::min(64.0, std::cbrt((double)std::max(T, 1u) / 8.0)));
  auto cell_of = [&](const V3& c, int* g) {
    const double ex[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z}, p[3] = {c.x - lo.x, c.y - lo.y, c.z - lo.z};
    for (int k = 0; k < 3; k++) g[k] = ex[k] > 0 ? std::min(G - 1, std::max(0, (int)(p[k] / ex[k] * G))) : 0;
  };
  std::vector<std::vector<uint32_t>> grid((size_t)G * G * G);
  for (uint32_t t = 0; t < T; t++) {
    int g[3];
    cell_of(cen[t], g);
    grid[((size_t)g[2] * G + g[1]) * G + g[0]].push_back(t);
  }
  std::vector<uint32_t> order;  // triangles in grid-cell order: the bounded fallback of the search below
  order.reserve(T);
  for (const auto& cell : grid) order.insert(order.end(), cell.begin(), cell.end());
  size_t cursor = 0;
  auto nearest_unused = [&](const V3& c) -> int64_t {
    int g[3];
    cell_of(c, g);
    int64_t best = -1;
    double bd = 1e300;
    for (int r = 0; r < std::min(G, 3); r++) {
      for (int z = std::max(0, g[2] - r); z <= std::min(G - 1, g[2] + r); z++)
        for (int y = std::max(0, g[1] - r); y <= std::min(G - 1, g[1] + r); y++)
          for (int x = std::max(0, g[0] - r); x <= std::min(G - 1, g[0] + r); x++) {
            if (std::max({std::abs(x - g[0]), std::abs(y - g[1]), std::abs(z - g[2])}) != r) continue;  // the shell only
            for (uint32_t t : grid[((size_t)z * G + y) * G + x]) {
              if (used[t]) continue;
              const V3 d = cen[t] - c;
              const double dd = dot(d, d);
              if (dd < bd || (dd == bd && (int64_t)t < best)) bd = dd, best = t;
            }
          }
      if (best >= 0 && r >= 1) break;  // one more shell than the first hit: the true nearest may sit in a neighbouring cell
    }
    if (best < 0) {  // nothing unused within two cells: the next unused triangle in cell order (O(T) over the whole build)
      while (cursor < order.size() && used[order[cursor]]) cursor++;
      if (cursor < order.size()) best = order[cursor];
    }
    return best;
  };

  std::vector<uint32_t> mverts;
  std::vector<uint8_t> mcorners;
  V3 msum = {0, 0, 0};
  auto flush = [&]() {
    if (mcorners.empty()) return;
    while (out.lti.size() % 4) out.lti.push_back(0);  // 4-byte aligned micro-index runs (AssetManager_GLTF.cpp:687)
    out.meshlets.push_back({(uint32_t)out.ivi.size(), (uint32_t)out.lti.size(), (uint32_t)mverts.size(), (uint32_t)(mcorners.size() / 3)});
    out.ivi.insert(out.ivi.end(), mverts.begin(), mverts.end());
    out.lti.insert(out.lti.end(), mcorners.begin(), mcorners.end());
    for (uint32_t v : mverts) slot[v] = 0xFF;
    mverts.clear();
    mcorners.clear();
    msum = {0, 0, 0};
  };
  auto extra_of = [&](uint32_t t) {
    uint32_t e = 0;
    for (int k = 0; k < 3; k++) e += slot[idx[t * 3 + k]] == 0xFF ? 1u : 0u;
    // corners that repeat inside one triangle cannot occur (degenerates were dropped by the caller)
    return e;
  };
  auto add = [&](uint32_t t) {
    for (int k = 0; k < 3; k++) {
      const uint32_t v = idx[t * 3 + k];
      if (slot[v] == 0xFF) {
        slot[v] = (uint8_t)mverts.size();
        mverts.push_back(v);
      }
      mcorners.push_back(slot[v]);
      live[v]--;
    }
    used[t] = 1;
    msum = {msum.x + cen[t].x, msum.y + cen[t].y, msum.z + cen[t].z};
  };
  uint32_t remaining = T;
  V3 last_centroid = T ? cen[0] : V3{0, 0, 0};
  while (remaining) {
    int64_t best = -1;
    uint32_t best_extra = 4, best_live = 0xFFFFFFFFu;
    double best_d = 1e300;
    const double ntri = (double)std::max<size_t>(mcorners.size() / 3, 1);
    const V3 c = mcorners.empty() ? last_centroid : V3{msum.x / ntri, msum.y / ntri, msum.z / ntri};
    for (uint32_t v : mverts)
      for (uint32_t a = off[v]; a < off[v + 1]; a++) {
        const uint32_t t = adj[a];
        if (used[t]) continue;
        const uint32_t e = extra_of(t);
        uint32_t lv = 0;
        for (int k = 0; k < 3; k++) lv += live[idx[t * 3 + k]];
        const V3 d = cen[t] - c;
        const double dd = dot(d, d);
        if (e < best_extra || (e == best_extra && (lv < best_live || (lv == best_live && (dd < best_d || (dd == best_d && (int64_t)t < best))))))
          best = t, best_extra = e, best_live = lv, best_d = dd;
      }
    if (best < 0) {
      best = nearest_unused(c);
      best_extra = extra_of((uint32_t)best);
    }
    if (mverts.size() + best_extra > max_v || mcorners.size() / 3 + 1 > max_t) {
      last_centroid = c;
      flush();
      continue;  // choose again for the fresh meshlet (its seed is the triangle nearest to the one just closed)
    }
    add((uint32_t)best);
    remaining--;
  }
  flush();
  while (out.lti.size() % 4) out.lti.push_back(0);
}
}  // namespace

struct oxc_mesh_build {
  std::vector<Lod> lods;
};

extern "C" {

oxc_status oxc_mesh_build_create(const oxc_mesh_build_desc* d, oxc_mesh_build** out) {
  if (!out) return OXC_INVALID_ARG;
  *out = nullptr;
  if (!d || d->struct_size != sizeof(oxc_mesh_build_desc) || !d->positions || !d->indices || d->index_count % 3 != 0) return OXC_INVALID_ARG;
  const uint32_t max_v = d->max_vertices ? d->max_vertices : 64u, max_t = d->max_triangles ? d->max_triangles : 64u;
  const uint32_t max_lods = d->max_lods ? std::min(d->max_lods, (uint32_t)OXC_MESH_MAX_LODS) : (uint32_t)OXC_MESH_MAX_LODS;
  if (max_v < 3 || max_v > 255 || max_t < 1 || max_t > 255) return OXC_INVALID_ARG;
  for (uint32_t i = 0; i < d->index_count; i++)
    if (d->indices[i] >= d->vertex_count) return OXC_INVALID_ARG;
  oxc_mesh_build* b = new (std::nothrow) oxc_mesh_build();
  if (!b) return OXC_OUT_OF_MEMORY;
  try {
    std::vector<uint32_t> last;  // the previous LOD's triangles without the degenerate ones (no area: neither clustered nor simplified)
    size_t last_count = 0;       // the previous LOD's index count as the reference sees it (LOD 0: the verbatim input)
    float last_error = 0.f;
    auto without_degenerates = [](const std::vector<uint32_t>& in) {
      std::vector<uint32_t> o;
      o.reserve(in.size());
      for (size_t i = 0; i + 2 < in.size(); i += 3)
        if (in[i] != in[i + 1] && in[i + 1] != in[i + 2] && in[i] != in[i + 2]) o.insert(o.end(), in.begin() + i, in.begin() + i + 3);
      return o;
    };
    for (uint32_t lod = 0; lod < max_lods; lod++) {  // AssetManager_GLTF.cpp:599-682
      Lod cur;
      if (lod == 0) {
        cur.indices.assign(d->indices, d->indices + d->index_count);  // LOD 0 = the input indices, verbatim (:604-606)
      } else {
        const size_t target = ((last_count + 5) / 6) * 3;  // :609
        Simplifier s;
        s.pos = d->positions;
        s.nrm = d->normals;
        s.vcount = d->vertex_count;
        s.init(last.data(), last.size());
        const double err = s.run(target);
        cur.indices = s.indices();
        cur.error = last_error + (float)err;  // :637
        if (cur.indices.size() > target + target / 2 || err > 0.5 || cur.indices.size() < 6) break;  // :639-645
      }
      if (cur.indices.size() < 3) break;
      last = without_degenerates(cur.indices);
      last_count = cur.indices.size();
      last_error = cur.error;
      build_meshlets(last, d->positions, d->vertex_count, max_v, max_t, cur);
      if (cur.meshlets.empty()) break;
      b->lods.push_back(std::move(cur));
    }
  } catch (const std::bad_alloc&) {
    delete b;
    return OXC_OUT_OF_MEMORY;
  }
  *out = b;
  return OXC_OK;
}

uint32_t oxc_mesh_build_lod_count(const oxc_mesh_build* b) { return b ? (uint32_t)b->lods.size() : 0u; }

oxc_status oxc_mesh_build_lod(const oxc_mesh_build* b, uint32_t lod, oxc_mesh_lod_view* out) {
  if (!b || !out || lod >= b->lods.size()) return OXC_INVALID_ARG;
  const Lod& l = b->lods[lod];
  out->indices = l.indices.data();
  out->indices_count = (uint32_t)l.indices.size();
  out->meshlets = l.meshlets.data();
  out->meshlet_count = (uint32_t)l.meshlets.size();
  out->indirect_vertex_indices = l.ivi.data();
  out->indirect_vertex_indices_count = (uint32_t)l.ivi.size();
  out->local_triangle_indices = l.lti.data();
  out->local_triangle_indices_count = (uint32_t)l.lti.size();
  out->error = l.error;
  return OXC_OK;
}

void oxc_mesh_build_destroy(oxc_mesh_build* b) { delete b; }

oxc_status oxc_mesh_vertex_fetch_remap(const uint32_t* stream, uint64_t count, uint32_t vertex_count, uint32_t* remap_out, uint32_t* used_out) {
  if ((!stream && count) || !remap_out) return OXC_INVALID_ARG;
  constexpr uint32_t kUnset = 0xFFFFFFFFu;
  for (uint32_t v = 0; v < vertex_count; v++) remap_out[v] = kUnset;
  uint32_t next = 0;
  for (uint64_t i = 0; i < count; i++) {
    const uint32_t v = stream[i];
    if (v >= vertex_count) return OXC_INVALID_ARG;
    if (remap_out[v] == kUnset) remap_out[v] = next++;
  }
  if (used_out) *used_out = next;
  for (uint32_t v = 0; v < vertex_count; v++)
    if (remap_out[v] == kUnset) remap_out[v] = next++;
  return OXC_OK;
}

}  // extern "C"
