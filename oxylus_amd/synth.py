"""Seeded synthetic scenes in the reference's GPU byte layouts.

Everything is a torch tensor so the same generator runs on the CPU (tests, oracle input) and
directly in HBM (benchmarks at 10M+ meshlets, no PCIe upload).  `Scene.bind()` patches the
64-bit pointer fields of GPU::Mesh / GPU::MeshLOD (SceneGPU.hpp:125-152) with the addresses of
the tensors on whatever device they live on -- host addresses for the CPU oracle, device VAs for
the HIP path.  Layouts: Oxylus/include/Scene/SceneGPU.hpp:84-152,222-229; producer conventions:
Oxylus/src/Asset/AssetManager_GLTF.cpp:590-761 (extent = full box size, u16x4 positions,
4-byte aligned micro-index runs).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from .lib import CullCamera


def _align(v: int, a: int) -> int:
    return (v + a - 1) // a * a


def perspective_reversed_z(fov_deg: float, aspect: float, near: float, far: float) -> torch.Tensor:
    """glm::perspective(radians(fov), aspect, far, near) with GLM_FORCE_DEPTH_ZERO_TO_ONE, then
    proj[1][1] *= -1 (Oxylus/src/Render/Camera.cpp:36-54).  Column-major 16 floats."""
    t = math.tan(math.radians(fov_deg) / 2.0)
    z_near, z_far = far, near  # swapped on purpose: reversed-Z
    m = [[0.0] * 4 for _ in range(4)]  # m[col][row]
    m[0][0] = 1.0 / (aspect * t)
    m[1][1] = -(1.0 / t)
    m[2][2] = z_far / (z_near - z_far)
    m[2][3] = -1.0
    m[3][2] = -(z_far * z_near) / (z_far - z_near)
    return torch.tensor([m[c][r] for c in range(4) for r in range(4)], dtype=torch.float32)


@dataclass
class SceneSpec:
    n_mesh_instances: int = 16
    meshlets_per_mesh: int = 64  # K: LOD-0 meshlets per mesh
    lod_count: int = 1
    verts_per_meshlet: int = 64
    tris_per_meshlet: int = 64
    with_geometry: bool = True
    ragged: bool = False       # random vertex/triangle counts per meshlet
    share_meshes: int = 0      # >0: that many distinct meshes shared by all instances (instanced variant)
    nonuniform_scale: bool = False
    seed: int = 0x0A1DE5
    scene_depth: float = 200.0  # instances fill x,y in [-D/2, D/2], z in [-D, 0.1 D]
    resolution: int = 4096
    fov_deg: float = 60.0
    near: float = 0.1
    far: float = 1000.0
    degenerate_cone_frac: float = 0.15


@dataclass
class Scene:
    spec: SceneSpec
    device: torch.device
    # blob arrays
    bounds: torch.Tensor          # int16 [B, 8]    GPU::MeshletBounds
    meshlets: torch.Tensor        # int32 [B, 4]    GPU::Meshlet
    micro: torch.Tensor           # uint8 [..]      local_triangle_indices (u8 stream)
    vidx: torch.Tensor            # int32 [..]      indirect_vertex_indices
    positions: torch.Tensor       # int16 [.., 4]   u16x4 half positions
    lods: torch.Tensor            # int64 [n_meshes*lod_count, 8]   GPU::MeshLOD
    meshes: torch.Tensor          # int64 [n_meshes, 8]             GPU::Mesh
    transforms: torch.Tensor      # float32 [M, 16]
    mesh_instances: torch.Tensor  # int32 [M, 5]
    meshlet_instances: torch.Tensor  # int32 [N, 2]  LOD-0 expansion (cull_meshes output for "all visible, LOD 0")
    camera: Dict[str, object] = field(default_factory=dict)
    # bookkeeping (host ints)
    n_meshes: int = 0
    lod_meshlet_counts: Optional[list] = None
    _lod_tables: Optional[dict] = None

    @property
    def n_mesh_instances(self) -> int:
        return int(self.mesh_instances.shape[0])

    @property
    def n_meshlet_instances(self) -> int:
        return int(self.meshlet_instances.shape[0])

    def cull_camera(self) -> CullCamera:
        cam = CullCamera()
        pv = self.camera["projection_view"]
        for i in range(16):
            cam.projection_view[i] = float(pv[i])
        for i in range(3):
            cam.position[i] = float(self.camera["position"][i])
        cam.acceptable_lod_error = float(self.camera["acceptable_lod_error"])
        cam.resolution[0] = float(self.camera["resolution"][0])
        cam.resolution[1] = float(self.camera["resolution"][1])
        cam.near_clip = float(self.camera["near_clip"])
        cam.mesh_instance_count = self.n_mesh_instances
        return cam

    def bind(self) -> "Scene":
        """(Re)write the pointer fields of meshes/lods for the current tensor addresses."""
        t = self._lod_tables
        dev = self.device
        lods = self.lods
        base_bounds, base_meshlets = self.bounds.data_ptr(), self.meshlets.data_ptr()
        base_micro, base_vidx, base_pos = self.micro.data_ptr(), self.vidx.data_ptr(), self.positions.data_ptr()
        lods[:, 0] = 0
        lods[:, 1] = base_meshlets + t["meshlet_start"].to(dev) * 16
        lods[:, 2] = base_bounds + t["meshlet_start"].to(dev) * 16
        lods[:, 3] = base_micro + t["micro_start"].to(dev)
        lods[:, 4] = base_vidx + t["vidx_start"].to(dev) * 4
        meshes = self.meshes
        meshes[:, 0] = base_pos + t["mesh_vertex_start"].to(dev) * 8
        meshes[:, 1] = 0
        meshes[:, 2] = 0
        L = self.spec.lod_count
        meshes[:, 4] = lods.data_ptr() + torch.arange(self.n_meshes, dtype=torch.int64, device=dev) * (64 * L)
        return self

    def to(self, device) -> "Scene":
        device = torch.device(device)
        kw = {}
        for name in ("bounds", "meshlets", "micro", "vidx", "positions", "lods", "meshes", "transforms",
                     "mesh_instances", "meshlet_instances"):
            kw[name] = getattr(self, name).to(device).contiguous().clone()
        s = Scene(spec=self.spec, device=device, camera=self.camera, n_meshes=self.n_meshes,
                  lod_meshlet_counts=self.lod_meshlet_counts, _lod_tables=self._lod_tables, **kw)
        return s.bind()

    def clone(self) -> "Scene":
        return self.to(self.device)

    def prefix(self, m: int, device=None) -> "Scene":
        """The sub-scene made of the first `m` mesh instances with everything they reference, copied to `device`
        (default: this scene's).  Needs one mesh per instance (share_meshes == 0): the blob arrays are mesh-major, so the
        sub-scene is a prefix of every array and all offsets stay valid.  bench.py uses it to hand the CPU checker a
        bounded sample of the very arrays the GPU culls."""
        assert self.spec.share_meshes == 0 and 0 < m <= self.n_mesh_instances
        device = torch.device(device) if device is not None else self.device
        L, K, t = self.spec.lod_count, self.spec.meshlets_per_mesh, self._lod_tables
        full = m == self.n_meshes

        def end(table, per, total):
            return int(total) if full else int(table[m * per].item())

        n_meshlets = end(t["meshlet_start"], L, self.bounds.shape[0])
        if self.spec.with_geometry:
            n_vidx, n_micro = end(t["vidx_start"], L, self.vidx.shape[0]), end(t["micro_start"], L, self.micro.shape[0])
            n_pos = end(t["mesh_vertex_start"], 1, self.positions.shape[0])
        else:
            n_vidx, n_micro, n_pos = self.vidx.shape[0], self.micro.shape[0], self.positions.shape[0]
        cp = lambda x: x.to(device).contiguous().clone()  # noqa: E731
        tables = {"meshlet_start": t["meshlet_start"][: m * L], "vidx_start": t["vidx_start"][: m * L], "micro_start": t["micro_start"][: m * L],
                  "mesh_vertex_start": t["mesh_vertex_start"][:m]}
        spec = SceneSpec(**{**self.spec.__dict__, "n_mesh_instances": m})
        s = Scene(spec=spec, device=device, bounds=cp(self.bounds[:n_meshlets]), meshlets=cp(self.meshlets[:n_meshlets]), micro=cp(self.micro[:n_micro]),
                  vidx=cp(self.vidx[:n_vidx]), positions=cp(self.positions[:n_pos]), lods=cp(self.lods[: m * L]), meshes=cp(self.meshes[:m]),
                  transforms=cp(self.transforms[:m]), mesh_instances=cp(self.mesh_instances[:m]), meshlet_instances=cp(self.meshlet_instances[: m * K]),
                  camera=self.camera, n_meshes=m, lod_meshlet_counts=self.lod_meshlet_counts, _lod_tables=tables)
        return s.bind()

    def slice(self, a: int, b: int, device=None) -> "Scene":
        """The sub-scene of mesh instances [a, b) with ONLY what they reference (a rank's bounds and geometry live on that rank
        alone), shard-local indices: mesh / transform indices, MeshletInstance records and visibility offsets are rebased to the range.
        Needs one mesh per instance (share_meshes == 0): the blob arrays are mesh-major, so the range is one run of every array."""
        assert self.spec.share_meshes == 0 and 0 <= a < b <= self.n_mesh_instances
        device = torch.device(device) if device is not None else self.device
        L, K, t = self.spec.lod_count, self.spec.meshlets_per_mesh, self._lod_tables
        mi = self.mesh_instances[a:b]
        idx = torch.arange(a, b, dtype=mi.dtype, device=mi.device)
        assert torch.equal(mi[:, 0], idx) and torch.equal(mi[:, 3], idx), "one mesh and one transform per instance, in instance order"

        def run(table, per, total):
            lo = int(table[a * per].item())
            hi = int(total) if b == self.n_meshes else int(table[b * per].item())
            return lo, hi

        m0, m1 = run(t["meshlet_start"], L, self.bounds.shape[0])
        if self.spec.with_geometry:
            v0, v1 = run(t["vidx_start"], L, self.vidx.shape[0])
            c0, c1 = run(t["micro_start"], L, self.micro.shape[0])
            p0, p1 = run(t["mesh_vertex_start"], 1, self.positions.shape[0])
        else:
            (v0, v1), (c0, c1), (p0, p1) = (0, self.vidx.shape[0]), (0, self.micro.shape[0]), (0, self.positions.shape[0])
        cp = lambda x: x.to(device).contiguous().clone()  # noqa: E731
        tables = {"meshlet_start": t["meshlet_start"][a * L: b * L] - m0, "vidx_start": t["vidx_start"][a * L: b * L] - v0,
                  "micro_start": t["micro_start"][a * L: b * L] - c0, "mesh_vertex_start": t["mesh_vertex_start"][a:b] - p0}
        mesh_instances = cp(mi)
        first_bit = int(mi[0, 4].item())
        mesh_instances[:, 0] -= a
        mesh_instances[:, 3] -= a
        mesh_instances[:, 4] -= first_bit  # visibility offsets: the shard's mask starts at its own bit 0
        lo_rec = int((self.meshlet_instances[:, 0] < a).sum().item())
        hi_rec = int((self.meshlet_instances[:, 0] < b).sum().item())
        mli = cp(self.meshlet_instances[lo_rec:hi_rec])
        mli[:, 0] -= a
        spec = SceneSpec(**{**self.spec.__dict__, "n_mesh_instances": b - a})
        s = Scene(spec=spec, device=device, bounds=cp(self.bounds[m0:m1]), meshlets=cp(self.meshlets[m0:m1]), micro=cp(self.micro[c0:c1]), vidx=cp(self.vidx[v0:v1]),
                  positions=cp(self.positions[p0:p1]), lods=cp(self.lods[a * L: b * L]), meshes=cp(self.meshes[a:b]), transforms=cp(self.transforms[a:b]),
                  mesh_instances=mesh_instances, meshlet_instances=mli, camera=self.camera, n_meshes=b - a, lod_meshlet_counts=self.lod_meshlet_counts,
                  _lod_tables=tables)
        return s.bind()

    def take(self, ranges, device=None) -> "Scene":
        """The sub-scene of several mesh-instance ranges [(a, b), ...] laid end to end (SURVEY 8e's interleaved assignment: a rank owns every
        world-th block of instances): slice() of each range, the pieces' arrays concatenated and their shard-local indices shifted by
        what lies in front of them -- mesh / transform indices by the instances, visibility offsets by the mask bits (K per instance),
        MeshletInstance records by the instances, the blob tables by the array runs.  One range gives slice(a, b)."""
        ranges = [(int(a), int(b)) for a, b in ranges if b > a]
        assert ranges, "at least one non-empty range"
        pieces = [self.slice(a, b, device) for a, b in ranges]
        if len(pieces) == 1:
            return pieces[0]
        K = self.spec.meshlets_per_mesh
        names = ("bounds", "meshlets", "micro", "vidx", "positions", "lods", "meshes", "transforms")
        run = {n: 0 for n in ("inst", "meshlet", "vidx", "micro", "vertex")}
        mis, mlis, tables = [], [], {k: [] for k in ("meshlet_start", "vidx_start", "micro_start", "mesh_vertex_start")}
        for pc in pieces:
            mi = pc.mesh_instances.clone()
            mi[:, 0] += run["inst"]
            mi[:, 3] += run["inst"]
            mi[:, 4] += run["inst"] * K
            mis.append(mi)
            ml = pc.meshlet_instances.clone()
            ml[:, 0] += run["inst"]
            mlis.append(ml)
            t = pc._lod_tables
            tables["meshlet_start"].append(t["meshlet_start"] + run["meshlet"])
            tables["vidx_start"].append(t["vidx_start"] + run["vidx"])
            tables["micro_start"].append(t["micro_start"] + run["micro"])
            tables["mesh_vertex_start"].append(t["mesh_vertex_start"] + run["vertex"])
            run["inst"] += pc.n_mesh_instances
            run["meshlet"] += pc.bounds.shape[0]
            run["vidx"] += pc.vidx.shape[0]
            run["micro"] += pc.micro.shape[0]
            run["vertex"] += pc.positions.shape[0]
        kw = {n: torch.cat([getattr(pc, n) for pc in pieces]).contiguous() for n in names}
        spec = SceneSpec(**{**self.spec.__dict__, "n_mesh_instances": run["inst"]})
        s = Scene(spec=spec, device=pieces[0].device, mesh_instances=torch.cat(mis).contiguous(), meshlet_instances=torch.cat(mlis).contiguous(),
                  camera=self.camera, n_meshes=run["inst"], lod_meshlet_counts=self.lod_meshlet_counts,
                  _lod_tables={k: torch.cat(v) for k, v in tables.items()}, **kw)
        return s.bind()

    def algorithmic_bytes_meshlet_stage(self, visible_fraction: float) -> float:
        """8 B MeshletInstance + 16 B MeshletBounds read, 4*v B written, per-mesh
        tables (20+64+64+64 B) amortised over K meshlets."""
        return 24.0 + 212.0 / max(1, self.spec.meshlets_per_mesh) + 4.0 * visible_fraction


def _f16_bits(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).view(torch.int16)


def _quat_to_mat(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r = torch.empty((q.shape[0], 3, 3), dtype=torch.float32, device=q.device)
    r[:, 0, 0] = 1 - 2 * (y * y + z * z)
    r[:, 0, 1] = 2 * (x * y - w * z)
    r[:, 0, 2] = 2 * (x * z + w * y)
    r[:, 1, 0] = 2 * (x * y + w * z)
    r[:, 1, 1] = 1 - 2 * (x * x + z * z)
    r[:, 1, 2] = 2 * (y * z - w * x)
    r[:, 2, 0] = 2 * (x * z - w * y)
    r[:, 2, 1] = 2 * (y * z + w * x)
    r[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return r


def make_scene(spec: SceneSpec, device="cpu", global_ids=None, global_total: int = 0) -> Scene:
    """global_ids / global_total (multi-GPU "one scene" form): this scene is the shard of a scene of `global_total` mesh instances
    that holds the instances with these global indices (int64 [n_mesh_instances]) -- they are PLACED where the whole scene's grid puts those
    indices (so the union of the ranks' shards is one spatially coherent scene and contiguous index ranges are slabs of it), everything else
    (geometry, rotation, scale, jitter) comes from this shard's own seed."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(spec.seed)

    def rand(*shape):
        return torch.rand(*shape, generator=g, device=dev, dtype=torch.float32)

    M, K, L = spec.n_mesh_instances, spec.meshlets_per_mesh, spec.lod_count
    V, T = spec.verts_per_meshlet, spec.tris_per_meshlet
    n_meshes = spec.share_meshes if spec.share_meshes > 0 else M
    lod_counts = [max(1, K >> i) for i in range(L)]
    per_mesh_meshlets = sum(lod_counts)
    B = n_meshes * per_mesh_meshlets
    mesh_half = 2.0  # meshlet centres uniform in [-2,2]^3 (mesh-local metres)

    # ---- meshlet bounds (all meshes, all lods; layout: mesh-major, then lod, then meshlet) ----
    centre = (rand(B, 3) * 2 - 1) * mesh_half
    extent = torch.exp(rand(B, 3) * math.log(10.0) + math.log(0.05))  # log-uniform 0.05..0.5
    axis = torch.randn(B, 3, generator=g, device=dev, dtype=torch.float32)
    axis = axis / axis.norm(dim=1, keepdim=True).clamp_min(1e-6)
    axis_s8 = torch.round(axis * 127).clamp(-127, 127).to(torch.int32)
    cutoff_s8 = torch.randint(-64, 128, (B,), generator=g, device=dev, dtype=torch.int32)
    cutoff_s8 = torch.where(rand(B) < spec.degenerate_cone_frac, torch.full_like(cutoff_s8, 127), cutoff_s8)
    bounds = torch.empty((B, 8), dtype=torch.int16, device=dev)
    bounds[:, 0:3] = _f16_bits(centre)
    bounds[:, 4:7] = _f16_bits(extent)
    bounds[:, 3] = ((axis_s8[:, 0] & 0xFF) | ((axis_s8[:, 1] & 0xFF) << 8)).to(torch.int16)
    bounds[:, 7] = ((axis_s8[:, 2] & 0xFF) | ((cutoff_s8 & 0xFF) << 8)).to(torch.int16)

    # ---- meshlet geometry ----
    lod_index_of = torch.arange(n_meshes * L, device=dev, dtype=torch.int64)
    lod_of = lod_index_of % L
    lod_cnt = torch.tensor(lod_counts, dtype=torch.int64, device=dev)[lod_of]          # meshlets per (mesh,lod)
    meshlet_start = torch.cumsum(lod_cnt, 0) - lod_cnt                                  # first meshlet of (mesh,lod)
    if spec.with_geometry:
        if spec.ragged:
            vcount = torch.randint(3, V + 1, (B,), generator=g, device=dev, dtype=torch.int64)
            tcount = torch.randint(1, T + 1, (B,), generator=g, device=dev, dtype=torch.int64)
        else:
            vcount = torch.full((B,), V, dtype=torch.int64, device=dev)
            tcount = torch.full((B,), T, dtype=torch.int64, device=dev)
        tbytes = (tcount * 3 + 3) // 4 * 4
        # offsets are relative to the (mesh,lod) arrays
        vstart_g = torch.cumsum(vcount, 0) - vcount
        tstart_g = torch.cumsum(tbytes, 0) - tbytes
        owner = torch.repeat_interleave(torch.arange(n_meshes * L, device=dev), lod_cnt)  # (mesh,lod) of each meshlet
        lod_vstart = vstart_g[meshlet_start]      # vertex start of each (mesh,lod)
        lod_tstart = tstart_g[meshlet_start]
        meshlets = torch.empty((B, 4), dtype=torch.int32, device=dev)
        meshlets[:, 0] = (vstart_g - lod_vstart[owner]).to(torch.int32)
        meshlets[:, 1] = (tstart_g - lod_tstart[owner]).to(torch.int32)
        meshlets[:, 2] = vcount.to(torch.int32)
        meshlets[:, 3] = tcount.to(torch.int32)
        total_v = int(vcount.sum().item())
        total_tb = int(tbytes.sum().item())
        # vertex positions: uniform inside the meshlet's AABB
        vert_owner = torch.repeat_interleave(torch.arange(B, device=dev), vcount)
        positions = torch.zeros((total_v, 4), dtype=torch.int16, device=dev)
        step = 1 << 24
        for s in range(0, total_v, step):
            o = vert_owner[s:s + step]
            p = centre[o] + (rand(o.shape[0], 3) - 0.5) * extent[o]
            positions[s:s + step, 0:3] = _f16_bits(p)
        # indirect vertex indices: index into the mesh's vertex array
        mesh_of_lod = lod_index_of // L
        mesh_vertex_start = lod_vstart[torch.arange(n_meshes, device=dev) * L]
        vidx = (torch.arange(total_v, device=dev, dtype=torch.int64) - mesh_vertex_start[mesh_of_lod[owner[vert_owner]]]).to(torch.int32)
        # micro indices: random corners < vertex_count
        micro = torch.zeros((total_tb,), dtype=torch.uint8, device=dev)
        byte_owner = torch.repeat_interleave(torch.arange(B, device=dev), tbytes)
        for s in range(0, total_tb, step):
            o = byte_owner[s:s + step]
            micro[s:s + step] = (rand(o.shape[0]) * vcount[o].to(torch.float32)).to(torch.int64).clamp_(max=255).to(torch.uint8)
        micro = torch.minimum(micro, (vcount[byte_owner] - 1).to(torch.uint8))
        vidx_start = lod_vstart
        micro_start = lod_tstart
        del vert_owner, byte_owner
    else:
        meshlets = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        positions = torch.zeros((1, 4), dtype=torch.int16, device=dev)
        vidx = torch.zeros((1,), dtype=torch.int32, device=dev)
        micro = torch.zeros((4,), dtype=torch.uint8, device=dev)
        vidx_start = torch.zeros(n_meshes * L, dtype=torch.int64, device=dev)
        micro_start = torch.zeros(n_meshes * L, dtype=torch.int64, device=dev)
        mesh_vertex_start = torch.zeros(n_meshes, dtype=torch.int64, device=dev)
        total_v, total_tb = 0, 0

    # ---- GPU::MeshLOD / GPU::Mesh tables (pointers filled by bind()) ----
    lods = torch.zeros((n_meshes * L, 8), dtype=torch.int64, device=dev)
    lods32 = lods.view(torch.int32)  # [.., 16]
    lods32[:, 10] = 0                                  # indices_count
    lods32[:, 11] = lod_cnt.to(torch.int32)            # meshlet_count
    lods32[:, 12] = lod_cnt.to(torch.int32)            # meshlet_bounds_count
    lods32[:, 13] = 0
    lods32[:, 14] = 0
    lodf = lod_of.to(torch.float32)
    err = torch.where(lod_of > 0, 0.004 * torch.pow(torch.full_like(lodf, 2.0), lodf), torch.zeros_like(lodf))
    lods32[:, 15] = err.to(torch.float32).view(torch.int32)
    meshes = torch.zeros((n_meshes, 8), dtype=torch.int64, device=dev)
    meshes32 = meshes.view(torch.int32)
    mesh_nverts = torch.zeros(n_meshes, dtype=torch.int64, device=dev)
    meshes32[:, 6] = mesh_nverts.to(torch.int32)
    meshes32[:, 7] = L
    mb_centre = torch.zeros((n_meshes, 3), dtype=torch.float32, device=dev)
    mb_extent = torch.full((n_meshes, 3), 2 * mesh_half + 0.5, dtype=torch.float32, device=dev)
    meshes32[:, 10:13] = mb_centre.view(torch.int32)
    meshes32[:, 13:16] = mb_extent.view(torch.int32)

    # ---- instances ----
    D = spec.scene_depth
    n_side = max(1, int(math.ceil((global_total if global_ids is not None else M) ** (1.0 / 3.0))))
    ids = torch.arange(M, device=dev, dtype=torch.int64)
    place = ids if global_ids is None else torch.as_tensor(global_ids, dtype=torch.int64, device=dev)
    assert place.numel() == M
    gx, gy, gz = place % n_side, (place // n_side) % n_side, place // (n_side * n_side)
    cell = torch.stack([gx, gy, gz], 1).to(torch.float32)
    jitter = rand(M, 3)
    u = (cell + jitter) / float(n_side)
    pos = torch.stack([(u[:, 0] - 0.5) * D, (u[:, 1] - 0.5) * D, -D + u[:, 2] * 1.1 * D], 1)
    q = torch.randn(M, 4, generator=g, device=dev, dtype=torch.float32)
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-6)
    R = _quat_to_mat(q)
    if spec.nonuniform_scale:
        S = 0.5 + 1.5 * rand(M, 3)
    else:
        S = (0.5 + 1.5 * rand(M, 1)).expand(M, 3)
    RS = R * S[:, None, :]
    world = torch.zeros((M, 4, 4), dtype=torch.float32, device=dev)  # [m, col, row]
    world[:, 0:3, 0:3] = RS.transpose(1, 2)
    world[:, 3, 0:3] = pos
    world[:, 3, 3] = 1.0
    transforms = world.reshape(M, 16).contiguous()

    mesh_index = (ids % n_meshes)
    mesh_instances = torch.zeros((M, 5), dtype=torch.int32, device=dev)
    mesh_instances[:, 0] = mesh_index.to(torch.int32)
    mesh_instances[:, 1] = 0
    mesh_instances[:, 2] = 0
    mesh_instances[:, 3] = ids.to(torch.int32)
    mesh_instances[:, 4] = (ids * K).to(torch.int32)  # running sum of LOD-0 meshlet counts (Scene.cpp:1248-1260)
    n_total = M * K
    mi = torch.arange(n_total, device=dev, dtype=torch.int64)
    meshlet_instances = torch.stack([(mi // K).to(torch.int32), (mi % K).to(torch.int32)], 1).contiguous()

    proj = perspective_reversed_z(spec.fov_deg, 1.0, spec.near, spec.far)
    camera = {
        "projection_view": proj.tolist(),  # view = identity: camera at the origin looking down -Z
        "position": [0.0, 0.0, 0.0],
        "acceptable_lod_error": 2.0,
        "resolution": [float(spec.resolution), float(spec.resolution)],
        "near_clip": spec.near,
    }
    tables = {
        "meshlet_start": meshlet_start.cpu(),
        "vidx_start": vidx_start.cpu(),
        "micro_start": micro_start.cpu(),
        "mesh_vertex_start": mesh_vertex_start.cpu(),
    }
    scene = Scene(spec=spec, device=dev, bounds=bounds, meshlets=meshlets, micro=micro, vidx=vidx, positions=positions,
                  lods=lods, meshes=meshes, transforms=transforms, mesh_instances=mesh_instances,
                  meshlet_instances=meshlet_instances, camera=camera, n_meshes=n_meshes,
                  lod_meshlet_counts=lod_counts, _lod_tables=tables)
    return scene.bind()


# ---------------------------------------------------------------------------------------------
# BASELINE configs[0]: ~1k entities, ECS transform update + host AABB frustum test (harness input)
# ---------------------------------------------------------------------------------------------
def camera_frustum_planes(position, forward, right, up, fov_deg: float, aspect: float, near: float, far: float):
    """Camera::get_frustum (Oxylus/src/Render/Camera.cpp:57-74) -> float32 [6, 4] {unit normal, distance = dot(normal, point)}
    (Frustum.hpp:7-18) in the order top, bottom, right, left, far, near."""
    import numpy as np

    p, f, r, u = (np.asarray(v, dtype=np.float64) for v in (position, forward, right, up))
    half_v = far * math.tan(math.radians(fov_deg) * 0.5)
    half_h = half_v * aspect
    ff = far * f
    faces = [(p, np.cross(r, ff - u * half_v)), (p, np.cross(ff + u * half_v, r)), (p, np.cross(ff - r * half_h, u)), (p, np.cross(u, ff + r * half_h)),
             (p + ff, -f), (p + near * f, f)]
    out = np.zeros((6, 4), dtype=np.float32)
    for i, (pt, n) in enumerate(faces):
        n = n / np.linalg.norm(n)
        out[i, :3], out[i, 3] = n, float(n @ pt)
    return out


def make_entities(n: int = 1000, depth: int = 3, seed: int = 0x0A1DE5 + 1):
    """`n` entities in parent chains of length `depth` (entity i's parent is i - 1 unless i starts a chain), each with a
    TransformComponent {translation, quaternion wxyz, scale} (Scene.cpp:1690-1711) and a baked AABB {min, max}.
    Returns (trs10 f32 [n,10], parent i32 [n], aabb6 f32 [n,6]) -- numpy, host memory (this path never touches the GPU)."""
    import numpy as np

    rng = np.random.default_rng(seed)
    trs = np.zeros((n, 10), dtype=np.float32)
    root = (np.arange(n) % depth) == 0
    trs[:, 0:3] = np.where(root[:, None], rng.uniform(-150.0, 150.0, (n, 3)) * np.array([1.0, 0.3, 1.0]), rng.uniform(-4.0, 4.0, (n, 3)))
    q = rng.standard_normal((n, 4))
    trs[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    trs[:, 7:10] = rng.uniform(0.5, 2.0, (n, 1))
    parent = np.where(root, -1, np.arange(n) - 1).astype(np.int32)
    half = rng.uniform(0.25, 2.0, (n, 3)).astype(np.float32)
    centre = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    aabb = np.concatenate([centre - half, centre + half], axis=1).astype(np.float32)
    return np.ascontiguousarray(trs), parent, np.ascontiguousarray(aabb)


# ---------------------------------------------------------------------------------------------
# HiZ helpers
# ---------------------------------------------------------------------------------------------
def hiz_extent_for(depth_w: int, depth_h: int):
    """RendererInstance.cpp:573-577: bit_ceil((dim + 1) >> 1) per axis."""
    def bit_ceil(v):
        return 1 << max(0, (v - 1).bit_length())
    return bit_ceil((depth_w + 1) >> 1), bit_ceil((depth_h + 1) >> 1)


def hiz_layout(w: int, h: int, levels: Optional[int] = None):
    """Level count min(floor(log2(max(w,h)))+1, 13) (RendererInstance.cpp:585, Texture.hpp:144-146)
    and a packed linear layout: byte offset of each level (256-byte aligned), total bytes."""
    if levels is None:
        levels = min(int(math.floor(math.log2(max(w, h)))) + 1, 13)
    offs, off = [], 0
    for k in range(levels):
        offs.append(off)
        off = _align(off + max(1, w >> k) * max(1, h >> k) * 4, 256)
    return levels, offs, off


def make_depth(w: int, h: int, n_quads: int = 64, seed: int = 1, device="cpu", near: float = 0.1) -> torch.Tensor:
    """Synthetic reversed-Z depth: far (0) background plus random screen-space quads at random
    view distances of 15..150 m (depth ~ near/dist), nearest wins."""
    dev = torch.device(device)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    depth = torch.zeros((h, w), dtype=torch.float32, device=dev)
    for _ in range(n_quads):
        cx, cy = torch.rand(2, generator=g).tolist()
        sx, sy = (0.03 + 0.22 * torch.rand(2, generator=g)).tolist()
        dist = 15.0 + 135.0 * float(torch.rand(1, generator=g))
        z = near / dist
        x0, x1 = int(max(0, (cx - sx / 2) * w)), int(min(w, (cx + sx / 2) * w))
        y0, y1 = int(max(0, (cy - sy / 2) * h)), int(min(h, (cy + sy / 2) * h))
        if x1 > x0 and y1 > y0:
            depth[y0:y1, x0:x1] = torch.clamp(depth[y0:y1, x0:x1], min=z)
    return depth


# ---------------------------------------------------------------------------------------------
# VSM (multi-view) helpers: Oxylus/src/Render/Passes/Shadowmaps.cpp:9-63
# ---------------------------------------------------------------------------------------------
def virtual_shadow_matrices(camera_position, light_dir, max_shadow_dist: float, first_clipmap_width: float, clipmap_count: int = 10,
                            page_table_size: int = 64):
    """calculate_virtual_shadow_matrices restated in float64 numpy (input generation only; the
    kernels take the resulting float32 matrices).  Returns (float32 [V,16] column-major
    projection_view_mat, int32 [V,2] page_offset, z_near)."""
    import numpy as np

    f = -np.asarray(light_dir, dtype=np.float64)
    f = f / np.linalg.norm(f)
    up = np.array([0.0, 1.0, 0.0])
    if 1.0 - abs(float(f @ up)) < 1e-5:
        up = np.array([0.0, 0.0, 1.0])
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    view = np.eye(4)
    view[0, :3], view[1, :3], view[2, :3] = s, u, -f  # lookAtRH from the origin
    mats, offs = [], []
    for i in range(clipmap_count):
        ext = first_clipmap_width * float(1 << i)
        n, fa = -max_shadow_dist, max_shadow_dist
        proj = np.eye(4)  # orthoRH_ZO(-ext, ext, -ext, ext, n, fa)
        proj[0, 0] = 1.0 / ext
        proj[1, 1] = -(1.0 / ext)  # proj[1][1] *= -1
        proj[2, 2] = -1.0 / (fa - n)
        proj[2, 3] = -n / (fa - n)
        clip = proj @ view @ np.append(np.asarray(camera_position, dtype=np.float64), 1.0)
        ndc = clip[:2] / clip[3]
        page_offset = np.trunc(ndc * 0.5 * page_table_size).astype(np.int32)
        shift = page_offset.astype(np.float64) / page_table_size * 2.0
        tr = np.eye(4)
        tr[0, 3], tr[1, 3] = -shift[0], -shift[1]
        final_view = np.linalg.inv(proj) @ (tr @ proj) @ view
        pvm = proj @ final_view
        mats.append(pvm.T.reshape(-1).astype(np.float32))  # column-major
        offs.append(page_offset)
    return np.stack(mats), np.stack(offs), float(-max_shadow_dist)


def pack_clipmaps(mats, offs, z_near: float) -> torch.Tensor:
    """-> uint8 [V*76]: GPU::VirtualClipmap records (SceneGPU.hpp:335-339)."""
    import numpy as np

    V = mats.shape[0]
    rec = np.zeros((V, 19), dtype=np.float32)
    rec[:, :16] = mats
    rec.view(np.int32)[:, 16:18] = offs
    rec[:, 18] = z_near
    return torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())


# ------------------------------------------------------------------------------------------
# Raw triangle meshes + a simple clusteriser: inputs for the meshlet bounds producer.
# The reference clusters with meshopt_buildMeshlets (AssetManager_GLTF.cpp:657-681); what the producer
# consumes is only that routine's OUTPUT FORMAT, restated here: per meshlet {vertex_offset, triangle_offset,
# vertex_count, triangle_count}, a u32 list of mesh vertex ids per meshlet and a u8 list of local corner
# indices whose per-meshlet start is 4-byte aligned.
# ------------------------------------------------------------------------------------------
def make_mesh(kind: str, n: int = 24, seed: int = 1):
    """Returns (positions f32 [V,3], triangles i64 [T,3]) on the CPU."""
    import numpy as np

    rng = np.random.default_rng(seed)
    if kind == "sphere":  # UV sphere: smooth normals, narrow cones
        lat, lon = np.meshgrid(np.linspace(0.05, np.pi - 0.05, n), np.linspace(0, 2 * np.pi, 2 * n, endpoint=False), indexing="ij")
        pos = np.stack([np.sin(lat) * np.cos(lon), np.cos(lat), np.sin(lat) * np.sin(lon)], -1).reshape(-1, 3) * 3.0 + np.array([1.0, -2.0, 0.5])
        rows, cols = n, 2 * n
        wrap = True
    elif kind in ("terrain", "plane0"):  # height field; "plane0": the x = +-0.0 plane (signed-zero folding)
        u, v = np.meshgrid(np.linspace(-4, 4, n), np.linspace(-4, 4, 2 * n), indexing="ij")
        if kind == "terrain":
            h = 0.6 * np.sin(1.7 * u) * np.cos(1.3 * v) + 0.15 * rng.standard_normal(u.shape)
            pos = np.stack([u, h, v], -1).reshape(-1, 3)
        else:
            zero = np.where(rng.random(u.shape) < 0.5, 0.0, -0.0)
            pos = np.stack([zero, u, v], -1).reshape(-1, 3)
        rows, cols = n, 2 * n
        wrap = False
    elif kind == "soup":  # unrelated triangles: cones wider than a hemisphere
        pos = rng.standard_normal((3 * n * n, 3)) * 2.0
        tris = np.arange(3 * n * n).reshape(-1, 3)
        return torch.from_numpy(pos.astype(np.float32)), torch.from_numpy(tris.astype(np.int64))
    else:
        raise ValueError(kind)
    tris = []
    for r in range(rows - 1):
        for c in range(cols - (0 if wrap else 1)):
            a, b = r * cols + c, r * cols + (c + 1) % cols
            d, e = (r + 1) * cols + c, (r + 1) * cols + (c + 1) % cols
            tris.append((a, d, b))
            tris.append((b, d, e))
    tris = np.asarray(tris, dtype=np.int64)
    if kind == "terrain":  # sprinkle degenerate triangles (repeated corner): left out of the cone, kept in the AABB
        k = rng.integers(0, len(tris), size=max(1, len(tris) // 40))
        tris[k, 2] = tris[k, 1]
    return torch.from_numpy(pos.astype(np.float32)), torch.from_numpy(tris)


def build_meshlets_simple(triangles: torch.Tensor, max_vertices: int = 64, max_triangles: int = 64):
    """Greedy in-order clusteriser.  Returns (meshlets i32 [M,4], vidx i32 [..], micro u8 [..])."""
    import numpy as np

    tris = triangles.numpy()
    meshlets, vidx, micro = [], [], []
    local, verts, corners = {}, [], []

    def flush():
        nonlocal local, verts, corners
        if not corners:
            return
        while len(micro) % 4:
            micro.append(0)
        meshlets.append((len(vidx), len(micro), len(verts), len(corners) // 3))
        vidx.extend(verts)
        micro.extend(corners)
        local, verts, corners = {}, [], []

    for t in tris:
        new = len({int(v) for v in t if int(v) not in local})
        if len(verts) + new > max_vertices or len(corners) // 3 + 1 > max_triangles:
            flush()
        for v in t:
            v = int(v)
            if v not in local:
                local[v] = len(verts)
                verts.append(v)
            corners.append(local[v])
    flush()
    while len(micro) % 4:
        micro.append(0)
    return (torch.tensor(meshlets, dtype=torch.int32).reshape(-1, 4), torch.tensor(vidx, dtype=torch.int32),
            torch.tensor(micro, dtype=torch.uint8))


def make_scene_from_mesh(n_mesh_instances: int, bounds: torch.Tensor, meshlets: torch.Tensor, micro: torch.Tensor, vidx: torch.Tensor,
                         positions_u16x4: torch.Tensor, mesh_bounds6: torch.Tensor, seed: int = 0x0A1DE5, device="cpu", **spec_kw) -> Scene:
    """A scene of `n_mesh_instances` randomly placed instances of ONE real mesh whose GPU arrays come from the
    asset path (clusteriser output + oxc_build_meshlet_bounds / its checker): the producer -> cull hand-over."""
    K = int(meshlets.shape[0])
    spec = SceneSpec(n_mesh_instances=n_mesh_instances, meshlets_per_mesh=K, share_meshes=1, with_geometry=False, seed=seed, **spec_kw)
    s = make_scene(spec, device)
    dev = s.device
    s.bounds = bounds.to(dev).contiguous().clone()
    s.meshlets = meshlets.to(dev).contiguous().clone()
    s.micro = micro.to(dev).contiguous().clone()
    s.vidx = vidx.to(dev).contiguous().clone()
    s.positions = positions_u16x4.to(dev).contiguous().clone()
    m32 = s.meshes.view(torch.int32)
    m32[0, 6] = int(positions_u16x4.shape[0])
    m32[0, 10:16] = mesh_bounds6.to(dev).to(torch.float32).view(torch.int32)
    z = torch.zeros(1, dtype=torch.int64)
    s._lod_tables = {"meshlet_start": z, "vidx_start": z, "micro_start": z, "mesh_vertex_start": z}
    s.spec = SceneSpec(**{**spec.__dict__, "with_geometry": True,
                          "tris_per_meshlet": int(meshlets[:, 3].max().item()) if K else 0,
                          "verts_per_meshlet": int(meshlets[:, 2].max().item()) if K else 0})
    return s.bind()
