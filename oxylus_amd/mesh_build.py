"""Host-side binding of the asset-side clusteriser (include/oxcull.h: oxc_mesh_build_*): triangle soup -> LOD chain -> meshlets,
the loop of Oxylus/src/Asset/AssetManager_GLTF.cpp:599-682.  Pure host code in the reference too; the GPU takes over at the bounds
producer (RendererInstance.build_meshlet_bounds / quantize_vertex_streams) and the mesh blob (oxylus_amd/mesh_blob.py)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from . import lib as L


def build_mesh_lods(positions: torch.Tensor, indices: torch.Tensor, normals: Optional[torch.Tensor] = None, max_lods: int = 0,
                    max_vertices: int = 64, max_triangles: int = 64) -> List[dict]:
    """positions f32 [V,3] (CPU), indices integer [T,3] or flat, normals f32 [V,3] or None.
    Returns one dict per LOD: indices i32 [n], meshlets i32 [M,4] (GPU::Meshlet), vidx i32 [..], micro u8 [..], error float."""
    lib = L.load()
    pos = np.ascontiguousarray(positions.detach().cpu().numpy(), dtype=np.float32)
    idx = np.ascontiguousarray(indices.detach().cpu().numpy().reshape(-1), dtype=np.uint32)
    nrm = np.ascontiguousarray(normals.detach().cpu().numpy(), dtype=np.float32) if normals is not None else None
    d = L.MeshBuildDesc()
    d.struct_size = C.sizeof(L.MeshBuildDesc)
    d.vertex_count, d.index_count, d.max_lods = pos.shape[0], idx.shape[0], max_lods
    d.max_vertices, d.max_triangles = max_vertices, max_triangles
    d.positions, d.indices = pos.ctypes.data, idx.ctypes.data
    d.normals = nrm.ctypes.data if nrm is not None else None
    h = C.c_void_p()
    st = lib.oxc_mesh_build_create(C.byref(d), C.byref(h))
    if st != L.OXC_OK:
        raise L.OxcError(st, "oxc_mesh_build_create: bad description (index out of range, index_count not a multiple of 3, limits outside 3..255)")
    try:
        out = []
        for i in range(lib.oxc_mesh_build_lod_count(h)):
            v = L.MeshLodView()
            st = lib.oxc_mesh_build_lod(h, i, C.byref(v))
            if st != L.OXC_OK:
                raise L.OxcError(st, "oxc_mesh_build_lod")

            def arr(ptr, n, ctype, dtype):
                return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)) if n else torch.zeros(0, dtype=getattr(torch, np.dtype(dtype).name))

            out.append({"indices": arr(v.indices, v.indices_count, C.c_uint32, np.int32),
                        "meshlets": arr(v.meshlets, v.meshlet_count * 4, C.c_uint32, np.int32).view(-1, 4),
                        "vidx": arr(v.indirect_vertex_indices, v.indirect_vertex_indices_count, C.c_uint32, np.int32),
                        "micro": arr(v.local_triangle_indices, v.local_triangle_indices_count, C.c_uint8, np.uint8), "error": float(v.error)})
        return out
    finally:
        lib.oxc_mesh_build_destroy(h)


def vertex_fetch_remap(first_use: torch.Tensor, vertex_count: int) -> torch.Tensor:
    """oxc_mesh_vertex_fetch_remap (include/oxcull.h; meshopt_optimizeVertexFetchRemap as AssetManager_GLTF.cpp:512-568 uses it): new id of
    every vertex = its rank by first appearance in `first_use` (an index stream); vertices the stream never names keep their relative order
    behind the used ones.  Returns remap[old] = new (int64)."""
    lib = L.load()
    ids = np.ascontiguousarray(first_use.detach().cpu().numpy().reshape(-1), dtype=np.uint32)
    remap = np.empty(vertex_count, dtype=np.uint32)
    used = C.c_uint32(0)
    st = lib.oxc_mesh_vertex_fetch_remap(ids.ctypes.data, ids.shape[0], vertex_count, remap.ctypes.data, C.byref(used))
    if st != L.OXC_OK:
        raise L.OxcError(st, "oxc_mesh_vertex_fetch_remap: an index >= vertex_count")
    return torch.from_numpy(remap.astype(np.int64))


def reorder_vertices(lods: List[dict], streams: List[torch.Tensor], by: str = "meshlets"):
    """Vertex order for the fetches of the triangle stage.  by="indices": first use in LOD 0's index buffer -- the reference's order
    (AssetManager_GLTF.cpp:512-568 remaps before it simplifies and clusters).  by="meshlets" (default): first use in LOD 0's
    indirect_vertex_indices, i.e. in meshlet order -- a meshlet's <= 64 vertices then sit next to each other in vertex_positions except
    for the ones an earlier meshlet introduced, and cull_triangles' position gather (64 lanes x 8 bytes) touches a handful of cache lines
    instead of one per grid row of the source mesh.  Same geometry, same meshlets, same bounds; only ids move.
    Returns (lods with remapped "vidx" / "indices", [stream[order] for stream in streams])."""
    V = int(streams[0].shape[0])
    remap = vertex_fetch_remap(lods[0]["vidx"] if by == "meshlets" else lods[0]["indices"], V)
    order = torch.empty(V, dtype=torch.int64)
    order[remap] = torch.arange(V, dtype=torch.int64)
    out = []
    for l in lods:
        m = dict(l)
        m["vidx"] = remap[l["vidx"].to(torch.int64)].to(torch.int32)
        m["indices"] = remap[l["indices"].to(torch.int64)].to(torch.int32)
        out.append(m)
    return out, [st[order.to(st.device)] for st in streams]


def make_scene_from_lods(n_mesh_instances: int, lods: List[dict], bounds_per_lod: List[torch.Tensor], positions_u16x4: torch.Tensor, mesh_bounds6: torch.Tensor,
                         seed: int = 0x0A1DE5, device="cpu", **spec_kw):
    """A scene of randomly placed instances of ONE mesh with its whole LOD chain (arrays LOD-major, as the mesh blob holds them):
    what cull_meshes' LOD select + expansion, cull_meshlets and cull_triangles consume."""
    from .synth import Scene, SceneSpec, make_scene

    Lc = len(lods)
    K0 = int(lods[0]["meshlets"].shape[0])
    spec = SceneSpec(n_mesh_instances=n_mesh_instances, meshlets_per_mesh=K0, share_meshes=1, lod_count=Lc, with_geometry=False, seed=seed, **spec_kw)
    s = make_scene(spec, device)
    dev = s.device
    cat = lambda key: torch.cat([l[key] for l in lods]).to(dev).contiguous()  # noqa: E731
    s.meshlets, s.vidx, s.micro = cat("meshlets"), cat("vidx"), cat("micro")
    s.bounds = torch.cat(list(bounds_per_lod)).to(dev).contiguous()
    s.positions = positions_u16x4.to(dev).contiguous().clone()
    counts = torch.tensor([int(l["meshlets"].shape[0]) for l in lods], dtype=torch.int64)
    start = lambda sizes: torch.cumsum(sizes, 0) - sizes  # noqa: E731
    s._lod_tables = {"meshlet_start": start(counts), "vidx_start": start(torch.tensor([int(l["vidx"].shape[0]) for l in lods], dtype=torch.int64)),
                     "micro_start": start(torch.tensor([int(l["micro"].shape[0]) for l in lods], dtype=torch.int64)), "mesh_vertex_start": torch.zeros(1, dtype=torch.int64)}
    l32 = s.lods.view(torch.int32)
    l32[:, 10] = torch.tensor([int(l["indices"].shape[0]) for l in lods], dtype=torch.int32)
    l32[:, 11] = counts.to(torch.int32)
    l32[:, 12] = counts.to(torch.int32)
    l32[:, 13] = torch.tensor([int(l["micro"].shape[0]) for l in lods], dtype=torch.int32)
    l32[:, 14] = torch.tensor([int(l["vidx"].shape[0]) for l in lods], dtype=torch.int32)
    l32[:, 15] = torch.tensor([l["error"] for l in lods], dtype=torch.float32).view(torch.int32)
    m32 = s.meshes.view(torch.int32)
    m32[0, 6] = int(positions_u16x4.shape[0])
    m32[0, 7] = Lc
    m32[0, 10:16] = mesh_bounds6.to(dev).to(torch.float32).view(torch.int32)
    s.lod_meshlet_counts = counts.tolist()
    s.spec = SceneSpec(**{**spec.__dict__, "with_geometry": True, "tris_per_meshlet": int(s.meshlets[:, 3].max().item()), "verts_per_meshlet": int(s.meshlets[:, 2].max().item())})
    return s.bind()


def make_scene_from_meshes(n_mesh_instances: int, meshes: List[dict], seed: int = 0x0A1DE5, device="cpu", **spec_kw):
    """A scene of randomly placed instances of SEVERAL built meshes (round robin), each with its own LOD chain.  `meshes`: one dict per
    mesh {"lods": build_mesh_lods(...), "bounds": [GPU::MeshletBounds tensor per LOD], "positions": u16x4 tensor, "mesh_bounds": 6 floats}.
    Arrays are mesh-major, then LOD-major (as one blob per mesh would hold them); meshes with a shorter chain leave their last
    GPU::MeshLOD rows empty (Mesh::lod_count says how many are valid).  meshlet_instances = the LOD-0 expansion of every instance,
    meshlet_instance_visibility_offset = running sum of the LOD-0 meshlet counts (Scene.cpp:1248-1260)."""
    from .synth import SceneSpec, make_scene

    n_meshes = len(meshes)
    Lmax = max(len(m["lods"]) for m in meshes)
    spec = SceneSpec(n_mesh_instances=n_mesh_instances, meshlets_per_mesh=1, share_meshes=n_meshes, lod_count=Lmax, with_geometry=False, seed=seed, **spec_kw)
    s = make_scene(spec, device)
    dev = s.device
    empty = {"meshlets": torch.zeros((0, 4), dtype=torch.int32), "vidx": torch.zeros(0, dtype=torch.int32), "micro": torch.zeros(0, dtype=torch.uint8),
             "indices": torch.zeros(0, dtype=torch.int32), "error": 0.0}
    rows = [(m["lods"][i] if i < len(m["lods"]) else empty, m["bounds"][i] if i < len(m["lods"]) else torch.zeros((0, 8), dtype=torch.int16))
            for m in meshes for i in range(Lmax)]
    s.meshlets = torch.cat([r[0]["meshlets"] for r in rows]).to(dev).contiguous()
    s.vidx = torch.cat([r[0]["vidx"] for r in rows]).to(dev).contiguous()
    s.micro = torch.cat([r[0]["micro"] for r in rows] + [torch.zeros(4, dtype=torch.uint8)]).to(dev).contiguous()
    s.bounds = torch.cat([r[1].cpu() for r in rows]).to(dev).contiguous()
    s.positions = torch.cat([m["positions"].cpu() for m in meshes]).to(dev).contiguous()
    sizes = lambda key: torch.tensor([int(r[0][key].shape[0]) for r in rows], dtype=torch.int64)  # noqa: E731
    start = lambda t: torch.cumsum(t, 0) - t  # noqa: E731
    counts = sizes("meshlets")
    nverts = torch.tensor([int(m["positions"].shape[0]) for m in meshes], dtype=torch.int64)
    s._lod_tables = {"meshlet_start": start(counts), "vidx_start": start(sizes("vidx")), "micro_start": start(sizes("micro")), "mesh_vertex_start": start(nverts)}
    l32 = s.lods.view(torch.int32)
    l32[:, 10] = sizes("indices").to(torch.int32)
    l32[:, 11] = counts.to(torch.int32)
    l32[:, 12] = counts.to(torch.int32)
    l32[:, 13] = sizes("micro").to(torch.int32)
    l32[:, 14] = sizes("vidx").to(torch.int32)
    l32[:, 15] = torch.tensor([r[0]["error"] for r in rows], dtype=torch.float32).view(torch.int32)
    m32 = s.meshes.view(torch.int32)
    m32[:, 6] = nverts.to(torch.int32)
    m32[:, 7] = torch.tensor([len(m["lods"]) for m in meshes], dtype=torch.int32)
    m32[:, 10:16] = torch.stack([m["mesh_bounds"].cpu().to(torch.float32).view(torch.int32) for m in meshes]).to(dev)
    k0 = counts.view(n_meshes, Lmax)[:, 0]  # LOD-0 meshlets per mesh
    per_inst = k0.to(dev)[s.mesh_instances[:, 0].to(torch.int64)]
    offs = torch.cumsum(per_inst, 0) - per_inst
    s.mesh_instances[:, 4] = offs.to(torch.int32)
    owner = torch.repeat_interleave(torch.arange(n_mesh_instances, device=dev, dtype=torch.int64), per_inst)
    k = torch.arange(int(per_inst.sum().item()), device=dev, dtype=torch.int64) - offs[owner]
    s.meshlet_instances = torch.stack([owner.to(torch.int32), k.to(torch.int32)], 1).contiguous()
    s.lod_meshlet_counts = None
    s.spec = SceneSpec(**{**spec.__dict__, "with_geometry": True, "tris_per_meshlet": int(s.meshlets[:, 3].max().item()), "verts_per_meshlet": int(s.meshlets[:, 2].max().item())})
    return s.bind()
