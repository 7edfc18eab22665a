"""Mesh blob + vertex streams (format side): the layout the cull kernels chase pointers into.
Host arithmetic (oxc_mesh_blob_layout_of / oxc_mesh_blob_finalize) runs without a GPU; the quantiser kernels and the
blob-backed frame are `gpu` tests.  Reference: Oxylus/src/Asset/AssetManager_GLTF.cpp:466-474, 570-597, 748-769, 773-800."""
import ctypes as C

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oxylus_amd import lib as L
from oxylus_amd.mesh_blob import MeshLodArrays, blob_desc, blob_layout, pack_mesh_blob
from oxylus_amd.synth import build_meshlets_simple, make_mesh, make_scene_from_mesh

LOD_FIELDS = ("indices", "meshlets", "meshlet_bounds", "local_triangle_indices", "indirect_vertex_indices")


def _desc(vertex_count, has_tex, counts):
    d = L.MeshBlobDesc()
    d.struct_size, d.vertex_count, d.has_texture_coords, d.lod_count = C.sizeof(L.MeshBlobDesc), vertex_count, int(has_tex), len(counts)
    for i, (ic, mc, lc, vc) in enumerate(counts[:L.MESH_MAX_LODS]):
        d.lods[i].indices_count, d.lods[i].meshlet_count = ic, mc
        d.lods[i].local_triangle_indices_count, d.lods[i].indirect_vertex_indices_count = lc, vc
    return d


def _as_dict(lay, n_lods):
    return {"vertex_positions": lay.vertex_positions, "vertex_normals": lay.vertex_normals, "texture_coords": lay.texture_coords,
            "lods": [{f: getattr(lay.lods[i], f) for f in LOD_FIELDS} for i in range(n_lods)],
            "lod_metadata_offset": lay.lod_metadata_offset, "size": lay.size}


def test_layout_known_answers(liboxcull):
    """Hand-derived from blob_append's rule (offset = align_up(size, alignment)):
    V=5 -> positions [0,40), normals [40,60); LOD0: indices 9*4 @ align8(60)=64 -> 100, meshlets 3*16 @ 104 -> 152,
    bounds 3*16 @ 152 -> 200, local 13 B @ 200 -> 213, indirect 7*4 @ align4(213)=216 -> 244, table @ align8(244)=248, +64."""
    lay = blob_layout(_desc(5, False, [(9, 3, 13, 7)]))
    assert _as_dict(lay, 1) == {"vertex_positions": 0, "vertex_normals": 40, "texture_coords": 0,
                                "lods": [{"indices": 64, "meshlets": 104, "meshlet_bounds": 152, "local_triangle_indices": 200,
                                          "indirect_vertex_indices": 216}], "lod_metadata_offset": 248, "size": 312}
    # with texcoords (3 vertices: 24 + 12 + 12 = 48) and two LODs; LOD1's arrays start where LOD0's indirect run ends
    lay = blob_layout(_desc(3, True, [(6, 1, 8, 4), (3, 1, 4, 3)]))
    want = {"vertex_positions": 0, "vertex_normals": 24, "texture_coords": 36,
            "lods": [{"indices": 48, "meshlets": 72, "meshlet_bounds": 88, "local_triangle_indices": 104, "indirect_vertex_indices": 112},
                     {"indices": 128, "meshlets": 144, "meshlet_bounds": 160, "local_triangle_indices": 176, "indirect_vertex_indices": 180}],
            "lod_metadata_offset": 192, "size": 192 + 128}
    assert _as_dict(lay, 2) == want


counts_st = st.tuples(st.integers(0, 3000), st.integers(0, 200), st.integers(0, 5000), st.integers(0, 3000))


@settings(max_examples=200, deadline=None)
@given(v=st.integers(0, 100000), tex=st.booleans(), counts=st.lists(counts_st, min_size=1, max_size=8))
def test_layout_matches_checker(liboxcull, oracle_lib, v, tex, counts):
    import oracle

    lay = blob_layout(_desc(v, tex, counts))
    got = _as_dict(lay, len(counts))
    assert got == oracle.mesh_blob_layout(v, tex, counts)
    # every array starts at its alignment and nothing overlaps the table
    for lod in got["lods"]:
        assert lod["indices"] % 8 == 0 and lod["meshlets"] % 8 == 0 and lod["meshlet_bounds"] % 8 == 0
        assert lod["local_triangle_indices"] % 8 == 0 and lod["indirect_vertex_indices"] % 4 == 0
    assert got["lod_metadata_offset"] % 8 == 0 and got["size"] == got["lod_metadata_offset"] + 64 * len(counts)


def test_layout_rejects_bad_descriptors(liboxcull):
    out = L.MeshBlobLayout()
    for n in (0, 9):
        d = _desc(4, False, [(3, 1, 4, 3)])
        d.lod_count = n
        assert liboxcull.oxc_mesh_blob_layout_of(C.byref(d), C.byref(out)) == L.OXC_INVALID_ARG
    d = _desc(4, False, [(3, 1, 4, 3)])
    d.struct_size -= 4
    assert liboxcull.oxc_mesh_blob_layout_of(C.byref(d), C.byref(out)) == L.OXC_INVALID_ARG
    assert liboxcull.oxc_mesh_blob_layout_of(None, C.byref(out)) == L.OXC_INVALID_ARG


def _two_lod_mesh(seed=5):
    pos, tris = make_mesh("sphere", n=12, seed=seed)
    lods = []
    for i, sub in enumerate((tris, tris[::2].contiguous())):  # LOD1: every other triangle (a stand-in for the simplifier's output)
        meshlets, vidx, micro = build_meshlets_simple(sub)
        lods.append((sub, meshlets, vidx, micro, 0.125 * i))
    return pos, lods


def test_finalize_relocates_like_upload(liboxcull, oracle_lib):
    """upload_gltf_mesh (AssetManager_GLTF.cpp:780-800): every offset + the device address, counts and error copied,
    texture_coords left 0 when absent, lods -> the table at the blob's tail."""
    import oracle

    pos, lods = _two_lod_mesh()
    qpos, qnrm, _ = oracle.quantize_vertex_streams(pos, torch.nn.functional.normalize(pos, dim=1), None)
    arrays = []
    for sub, meshlets, vidx, micro, err in lods:
        b, _, _ = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
        arrays.append(MeshLodArrays(sub.reshape(-1).to(torch.int32), meshlets, b, micro, vidx, err))
    desc = blob_desc(pos.shape[0], False, arrays)
    lay = blob_layout(desc)
    host = torch.zeros(lay.size, dtype=torch.uint8)
    mesh = torch.zeros(8, dtype=torch.int64)
    base = 0x7F12_3456_7000
    b6 = (C.c_float * 6)(1, 2, 3, 4, 5, 6)
    assert liboxcull.oxc_mesh_blob_finalize(C.byref(desc), C.byref(lay), base, host.data_ptr(), lay.size, C.byref(b6), mesh.data_ptr()) == L.OXC_OK
    assert mesh[0].item() == base + lay.vertex_positions and mesh[1].item() == base + lay.vertex_normals and mesh[2].item() == 0
    assert mesh.view(torch.int32)[6].item() == pos.shape[0] and mesh.view(torch.int32)[7].item() == 2
    assert mesh[4].item() == base + lay.lod_metadata_offset
    assert mesh.view(torch.float32)[10:16].tolist() == [1, 2, 3, 4, 5, 6]
    table = host[lay.lod_metadata_offset:].view(torch.int64).reshape(2, 8)
    for i, a in enumerate(arrays):
        for k, f in enumerate(LOD_FIELDS):
            assert table[i, k].item() == base + getattr(lay.lods[i], f)
        c32 = table[i].view(torch.int32)
        assert c32[10:15].tolist() == [a.indices.numel(), a.meshlets.shape[0], a.meshlets.shape[0], a.local_triangle_indices.numel(),
                                       a.indirect_vertex_indices.numel()]
        assert table[i].view(torch.float32)[15].item() == np.float32(a.error)
    # too-small blob / null arguments are refused, nothing written
    assert liboxcull.oxc_mesh_blob_finalize(C.byref(desc), C.byref(lay), base, host.data_ptr(), lay.size - 1, C.byref(b6), mesh.data_ptr()) == L.OXC_INVALID_ARG
    assert liboxcull.oxc_mesh_blob_finalize(C.byref(desc), C.byref(lay), base, None, lay.size, C.byref(b6), mesh.data_ptr()) == L.OXC_INVALID_ARG


def _blob_scene(device, n_lods=1):
    """One real mesh packed into a blob on `device` (n_lods of its two LODs); returns (scene whose GPU::Mesh points into
    the blob, the plain-array LOD-0 scene -- what a one-LOD blob must behave like)."""
    import oracle

    pos, lods = _two_lod_mesh(seed=9)
    nrm = torch.nn.functional.normalize(pos, dim=1)
    qpos, qnrm, _ = oracle.quantize_vertex_streams(pos, nrm, None)
    arrays, mesh6 = [], None
    for sub, meshlets, vidx, micro, err in lods:
        b, m6, _ = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
        mesh6 = m6 if mesh6 is None else mesh6  # gpu_mesh.bounds is LOD 0's (AssetManager_GLTF.cpp:739-746)
        arrays.append(MeshLodArrays(sub.reshape(-1).to(torch.int32), meshlets, b, micro, vidx, err * 0.024))  # LOD 1: error 0.003
    a0 = arrays[0]
    plain = make_scene_from_mesh(24, a0.meshlet_bounds, a0.meshlets, a0.local_triangle_indices, a0.indirect_vertex_indices, qpos, mesh6,
                                 seed=31, device="cpu", scene_depth=40.0)
    blob, mesh, lay = pack_mesh_blob(qpos, qnrm, None, arrays[:n_lods], mesh6, device)
    scene = plain.to(device)
    scene.meshes = mesh.view(1, 8).to(device)  # no bind() from here on: the record points into the blob
    scene._blob = blob
    return scene, plain


def test_blob_backed_mesh_drives_the_checker_like_plain_arrays(liboxcull, oracle_lib):
    """The packed blob is a drop-in for the separate arrays: the CPU checker, chasing GPU::Mesh -> lods -> arrays through the
    blob, produces the frame it produces from the plain arrays."""
    from util import assert_same, oracle_frame

    scene, plain = _blob_scene("cpu")
    want = oracle_frame(plain, run_cull_meshes=True)
    got = oracle_frame(scene, run_cull_meshes=True)
    assert_same(want, got, ["total", "visible", "indices"])
    assert 0 < len(want["visible"]) < want["total"]


@pytest.mark.gpu
def test_gpu_quantize_vertex_streams_matches_checker(renderer, oracle_lib):
    import oracle

    g = torch.Generator().manual_seed(404)
    V = 100_003
    pos = (torch.rand((V, 3), generator=g) * 2 - 1) * torch.tensor([1e-6, 1.0, 7e4])[torch.randint(0, 3, (V, 1), generator=g)]
    nrm = torch.nn.functional.normalize(torch.randn((V, 3), generator=g), dim=1)
    uv = torch.rand((V, 2), generator=g) * 4 - 1
    specials = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.5, -1.5, float("inf"), float("-inf"), float("nan"), 65504.0, 65520.0, 6.1e-5, 5.96e-8,
                             0.5 / 511, 1.5 / 511, -0.5 / 511, 1e-40])
    for t in (pos, nrm, uv):
        flat = t.reshape(-1)
        flat[:specials.numel()] = specials
    want = oracle.quantize_vertex_streams(pos, nrm, uv)
    got = renderer.quantize_vertex_streams(pos.cuda(), nrm.cuda(), uv.cuda())
    torch.cuda.synchronize()
    for w, q, name in zip(want, got, ("positions", "normals", "texcoords")):
        assert torch.equal(w, q.cpu()), name
    # absent streams are skipped and their outputs untouched
    only_n = renderer.quantize_vertex_streams(None, nrm.cuda(), None)
    assert only_n[0] is None and only_n[2] is None and torch.equal(only_n[1].cpu(), want[1])


@pytest.mark.gpu
def test_gpu_blob_backed_frame_matches_checker(renderer, oracle_lib):
    """cull_meshes + cull_meshlets + cull_triangles with the mesh resident as ONE blob (GPU::Mesh / GPU::MeshLOD holding absolute
    device addresses written by oxc_mesh_blob_finalize) == the checker on the plain arrays."""
    from util import assert_same, gpu_frame, oracle_frame

    scene, plain = _blob_scene("cuda")
    want = oracle_frame(plain, run_cull_meshes=True)
    got = gpu_frame(renderer, scene, run_cull_meshes=True)
    assert_same(want, got, ["total", "visible", "indices"])


@pytest.mark.gpu
def test_gpu_two_lod_blob_selects_and_culls_like_checker(renderer, oracle_lib):
    """Two LODs in one blob: cull_meshes walks GPU::Mesh::lods (the table oxc_mesh_blob_finalize wrote at the blob's tail), picks
    LOD 1 for the far instances, and the expansion / meshlet / triangle stages read that LOD's arrays -- same frame from the
    checker chasing the host copy of the same blob."""
    from util import assert_same, gpu_frame, oracle_frame

    host_scene, _ = _blob_scene("cpu", n_lods=2)
    gpu_scene, _ = _blob_scene("cuda", n_lods=2)
    want = oracle_frame(host_scene, run_cull_meshes=True)
    assert set(np.unique(want["lod_index"]).tolist()) == {0, 1}
    got = gpu_frame(renderer, gpu_scene, run_cull_meshes=True)
    assert_same(want, got, ["total", "lod_index", "visible", "indices"])


def test_packed_blob_round_trips_every_array(liboxcull, oracle_lib):
    """Every array can be read back from the packed blob at the offsets the LOD table / GPU::Mesh record point to (here: host
    addresses), texcoords included; padding bytes between arrays are zero."""
    import oracle

    pos, lods = _two_lod_mesh(seed=3)
    nrm = torch.nn.functional.normalize(pos, dim=1)
    uv = pos[:, :2] * 0.25 + 0.5
    qpos, qnrm, quv = oracle.quantize_vertex_streams(pos, nrm, uv)
    arrays = []
    for sub, meshlets, vidx, micro, err in lods:
        b, m6, _ = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
        arrays.append(MeshLodArrays(sub.reshape(-1).to(torch.int32), meshlets, b, micro, vidx, err))
    blob, mesh, lay = pack_mesh_blob(qpos, qnrm, quv, arrays, torch.tensor([0., 0, 0, 1, 1, 1]), "cpu")
    base = blob.data_ptr()
    used = torch.zeros(lay.size, dtype=torch.bool)

    def view(addr, like):
        off, n = addr - base, like.numel() * like.element_size()
        assert 0 <= off and off + n <= lay.size and not used[off:off + n].any()  # in range, no overlap
        used[off:off + n] = True
        return blob[off:off + n].view(like.dtype).reshape(like.shape)

    assert torch.equal(view(mesh[0].item(), qpos), qpos)
    assert torch.equal(view(mesh[1].item(), qnrm), qnrm)
    assert torch.equal(view(mesh[2].item(), quv), quv)
    table = view(mesh[4].item(), torch.zeros((2, 8), dtype=torch.int64)).clone()
    for i, a in enumerate(arrays):
        for k, t in enumerate((a.indices, a.meshlets, a.meshlet_bounds, a.local_triangle_indices, a.indirect_vertex_indices)):
            assert torch.equal(view(table[i, k].item(), t), t)
    assert not blob[~used].any()  # alignment padding only
