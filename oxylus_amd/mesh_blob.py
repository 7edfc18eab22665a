"""Host side of the mesh blob (format side): the one-allocation-per-mesh layout the cull kernels chase
pointers into.  Mirrors build_gltf_mesh's blob_append sequence and upload_gltf_mesh's relocation
(Oxylus/src/Asset/AssetManager_GLTF.cpp:466-474, 590-597, 748-769, 773-818) on top of the C ABI
(oxc_mesh_blob_layout_of / oxc_mesh_blob_finalize); the arrays themselves come from the clusteriser and from
oxc_build_meshlet_bounds / oxc_quantize_vertex_streams."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import lib as L


@dataclass
class MeshLodArrays:
    """One LOD's five arrays (GPU::MeshLOD, SceneGPU.hpp:125-139)."""
    indices: torch.Tensor                  # int32 [I]      simplified index buffer
    meshlets: torch.Tensor                 # int32 [M, 4]   GPU::Meshlet
    meshlet_bounds: torch.Tensor           # int16 [M, 8]   GPU::MeshletBounds
    local_triangle_indices: torch.Tensor   # uint8 [..]
    indirect_vertex_indices: torch.Tensor  # int32 [..]
    error: float = 0.0


def blob_desc(vertex_count: int, has_texture_coords: bool, lods: List[MeshLodArrays]) -> L.MeshBlobDesc:
    d = L.MeshBlobDesc()
    d.struct_size = C.sizeof(L.MeshBlobDesc)
    d.vertex_count, d.has_texture_coords, d.lod_count = int(vertex_count), int(bool(has_texture_coords)), len(lods)
    for i, lod in enumerate(lods[:L.MESH_MAX_LODS]):
        c = d.lods[i]
        c.indices_count = lod.indices.numel()
        c.meshlet_count = lod.meshlets.shape[0]
        c.local_triangle_indices_count = lod.local_triangle_indices.numel()
        c.indirect_vertex_indices_count = lod.indirect_vertex_indices.numel()
        c.error = float(lod.error)
    return d


def blob_layout(desc: L.MeshBlobDesc) -> L.MeshBlobLayout:
    out = L.MeshBlobLayout()
    st = L.load().oxc_mesh_blob_layout_of(C.byref(desc), C.byref(out))
    if st != 0:
        raise L.OxcError(st, "oxc_mesh_blob_layout_of: bad descriptor (lod_count must be 1..8)")
    return out


def pack_mesh_blob(quantized_positions: torch.Tensor, quantized_normals: torch.Tensor, quantized_texcoords: Optional[torch.Tensor],
                   lods: List[MeshLodArrays], mesh_bounds6: torch.Tensor, device) -> Tuple[torch.Tensor, torch.Tensor, L.MeshBlobLayout]:
    """-> (blob uint8 [size] on `device`, GPU::Mesh record as int64 [8] on the host, layout).
    The blob holds absolute addresses of itself (the LOD table), so it must not be moved afterwards."""
    V = quantized_positions.shape[0]
    assert quantized_normals.numel() == V and (quantized_texcoords is None or quantized_texcoords.shape[0] == V)
    desc = blob_desc(V, quantized_texcoords is not None, lods)
    lay = blob_layout(desc)
    host = torch.zeros(lay.size, dtype=torch.uint8)

    def put(offset: int, t: torch.Tensor):
        raw = t.detach().cpu().contiguous().view(torch.uint8).reshape(-1)
        host[offset:offset + raw.numel()] = raw

    put(lay.vertex_positions, quantized_positions)
    put(lay.vertex_normals, quantized_normals)
    if quantized_texcoords is not None:
        put(lay.texture_coords, quantized_texcoords)
    for i, lod in enumerate(lods):
        o = lay.lods[i]
        put(o.indices, lod.indices)
        put(o.meshlets, lod.meshlets)
        put(o.meshlet_bounds, lod.meshlet_bounds)
        put(o.local_triangle_indices, lod.local_triangle_indices)
        put(o.indirect_vertex_indices, lod.indirect_vertex_indices)
    device = torch.device(device)
    blob = host if device.type == "cpu" else torch.empty(lay.size, dtype=torch.uint8, device=device)
    mesh = torch.zeros(8, dtype=torch.int64)
    b6 = (C.c_float * 6)(*[float(x) for x in mesh_bounds6.detach().cpu().reshape(-1).tolist()])
    st = L.load().oxc_mesh_blob_finalize(C.byref(desc), C.byref(lay), C.c_uint64(blob.data_ptr()), C.c_void_p(host.data_ptr()), C.c_uint64(lay.size),
                                         C.byref(b6), C.c_void_p(mesh.data_ptr()))
    if st != 0:
        raise L.OxcError(st, "oxc_mesh_blob_finalize: bad arguments")
    if blob is not host:
        blob.copy_(host)
    return blob, mesh, lay
