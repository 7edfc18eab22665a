"""ctypes binding of liboxcull.so (the C ABI in include/oxcull.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this
module raises, it never routes to oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboxcull.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

OXC_OK, OXC_INVALID_ARG, OXC_HIP_ERROR, OXC_RCCL_ERROR, OXC_OUT_OF_MEMORY = range(5)
ABI_VERSION = 5  # OXC_ABI_VERSION of include/oxcull.h

CULL_TEST_FRUSTUM = 1
CULL_SELECT_LOD = 2
CULL_TEST_OCCLUSION = 4
CULL_LATE_PASS = 8
CULL_TEST_ALL = 7

STAGE_MESHES = 1
STAGE_MESHLETS = 2
STAGE_TRIANGLES = 4
STAGE_ALL = 7

TUNE_ASYNC_MTEST_BLOCKS_PER_CU, TUNE_ASYNC_TRI_BLOCKS_PER_CU, TUNE_RASTER_BIG_CAPACITY, TUNE_TRI_BLOCKS_PER_CU, TUNE_MV_EXPAND_ASYNC, TUNE_TRI_LOADS = 0, 1, 2, 3, 5, 7  # oxc_debug_set_tuning knobs


class Buffer(C.Structure):
    _fields_ = [("dptr", C.c_void_p), ("bytes", C.c_uint64)]


class Image(C.Structure):
    _fields_ = [
        ("dptr", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("levels", C.c_uint32),
        ("_pad", C.c_uint32),
        ("level_offset", C.c_uint64 * 13),
    ]


class ImageArrayU8(C.Structure):
    _fields_ = [
        ("dptr", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("layers", C.c_uint32),
        ("levels", C.c_uint32),
        ("level_offset", C.c_uint64 * 13),
    ]


class VirtualClipmap(C.Structure):
    _fields_ = [("projection_view_mat", C.c_float * 16), ("page_offset", C.c_int32 * 2), ("z_near", C.c_float)]


class CullCamera(C.Structure):
    _fields_ = [
        ("projection_view", C.c_float * 16),
        ("position", C.c_float * 3),
        ("acceptable_lod_error", C.c_float),
        ("resolution", C.c_float * 2),
        ("near_clip", C.c_float),
        ("mesh_instance_count", C.c_uint32),
    ]


class PreparedFrame(C.Structure):
    _fields_ = [
        ("mesh_instance_count", C.c_uint32),
        ("max_meshlet_instance_count", C.c_uint32),
        ("meshes_buffer", Buffer),
        ("transforms_world_buffer", Buffer),
        ("mesh_instances_buffer", Buffer),
        ("meshlet_instances_buffer", Buffer),
        ("visible_meshlet_instances_indices_buffer", Buffer),
        ("meshlet_instance_visibility_mask_buffer", Buffer),
        ("reordered_indices_buffer", Buffer),
    ]


class CullGeometryContext(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("use_hiz", C.c_uint32),
        ("use_hpb", C.c_uint32),
        ("init_cull_meshes", C.c_uint32),
        ("cull_flags", C.c_uint32),
        ("stages", C.c_uint32),
        ("cull_camera", CullCamera),
        ("hiz_attachment", Image),
        ("hpb_attachment", ImageArrayU8),
        ("vsm_clipmaps_buffer", Buffer),
        ("vsm_clipmap_dirty_flags_buffer", Buffer),
        ("vsm_clipmap_count", C.c_uint32),
        ("wide_triangle_index", C.c_uint32),
        ("small_triangle_cull", C.c_uint32),
        ("async_triangles", C.c_uint32),
        ("share_pass_tests", C.c_uint32),
        ("unordered_output", C.c_uint32),
        ("implicit_meshlet_instances", C.c_uint32),
        ("_reserved1", C.c_uint32),
        ("visibility_buffer", Buffer),
        ("cull_meshlets_cmd_buffer", Buffer),
        ("cull_triangles_cmd_buffer", Buffer),
        ("draw_geometry_cmd_buffer", Buffer),
        ("meshlet_instance_runs_buffer", Buffer),
    ]


class MainGeometryContext(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("_pad", C.c_uint32),
        ("depth_attachment", Image),
        ("hiz_attachment", Image),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("total_visible_meshlet_instances", C.c_uint32),
        ("early_visible_meshlet_instances", C.c_uint32),
        ("late_visible_meshlet_instances", C.c_uint32),
        ("cull_meshlets_cmd_x", C.c_uint32),
        ("cull_triangles_cmd_x", C.c_uint32),
        ("draw_index_count", C.c_uint32),
    ]


class MeshletBoundsDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("meshlet_count", C.c_uint32),
        ("_pad", C.c_uint32),
        ("positions", Buffer),
        ("meshlets", Buffer),
        ("indirect_vertex_indices", Buffer),
        ("local_triangle_indices", Buffer),
        ("meshlet_bounds", Buffer),
        ("mesh_bounds", Buffer),
        ("quantized_positions", Buffer),
    ]


class VertexStreamsDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("positions", Buffer),
        ("normals", Buffer),
        ("texcoords", Buffer),
        ("quantized_positions", Buffer),
        ("quantized_normals", Buffer),
        ("quantized_texcoords", Buffer),
    ]


MESH_MAX_LODS = 8


class MeshLodCounts(C.Structure):
    _fields_ = [
        ("indices_count", C.c_uint32),
        ("meshlet_count", C.c_uint32),
        ("local_triangle_indices_count", C.c_uint32),
        ("indirect_vertex_indices_count", C.c_uint32),
        ("error", C.c_float),
    ]


class MeshBlobDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("has_texture_coords", C.c_uint32),
        ("lod_count", C.c_uint32),
        ("lods", MeshLodCounts * MESH_MAX_LODS),
    ]


class MeshLodOffsets(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("indices", "meshlets", "meshlet_bounds", "local_triangle_indices", "indirect_vertex_indices")]


class MeshBlobLayout(C.Structure):
    _fields_ = [
        ("size", C.c_uint64),
        ("lod_metadata_offset", C.c_uint64),
        ("vertex_positions", C.c_uint64),
        ("vertex_normals", C.c_uint64),
        ("texture_coords", C.c_uint64),
        ("lods", MeshLodOffsets * MESH_MAX_LODS),
    ]


class TerrainContext(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("cull_flags", C.c_uint32),
        ("cull_camera", CullCamera),
        ("world_min", C.c_float * 2),
        ("world_size", C.c_float * 2),
        ("patch_count", C.c_uint32 * 2),
        ("base_height", C.c_float),
        ("height_scale", C.c_float),
        ("patch_minmax_attachment", Image),
        ("hiz_attachment", Image),
        ("visible_patches_buffer", Buffer),
        ("patch_visibility_mask_buffer", Buffer),
        ("draw_cmd_buffer", Buffer),
    ]


class DrawContext(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("wide_triangle_index", C.c_uint32),
        ("clear", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("_pad", C.c_uint32),
        ("projection_view", C.c_float * 16),
        ("draw_geometry_cmd_buffer", Buffer),
        ("visdepth_buffer", Buffer),
        ("depth_attachment", Image),
        ("visbuffer_attachment", Buffer),
    ]


# every symbol include/oxcull.h declares
EXPORTS = [
    "oxc_abi_version",
    "oxc_create",
    "oxc_destroy",
    "oxc_last_error",
    "oxc_reserve",
    "oxc_generate_hiz",
    "oxc_cull_geometry",
    "oxc_cull_geometry_batch",
    "oxc_join_triangles",
    "oxc_seed_meshlet_instances",
    "oxc_read_counters",
    "oxc_stream_read_probe",
    "oxc_debug_decode_bounds",
    "oxc_debug_raster_stats",
    "oxc_profile_begin",
    "oxc_profile_end",
    "oxc_build_meshlet_bounds",
    "oxc_quantize_vertex_streams",
    "oxc_mesh_blob_layout_of",
    "oxc_mesh_blob_finalize",
    "oxc_mesh_build_create",
    "oxc_mesh_build_lod_count",
    "oxc_mesh_build_lod",
    "oxc_mesh_build_destroy",
    "oxc_mesh_vertex_fetch_remap",
    "oxc_generate_hpb",
    "oxc_cull_terrain",
    "oxc_draw_visbuffer",
    "oxc_comm_unique_id",
    "oxc_comm_init",
    "oxc_comm_destroy",
    "oxc_pack_counters",
    "oxc_pack_counters_batch",
    "oxc_exchange_counts",
    "oxc_broadcast_hiz",
    "oxc_broadcast_hiz_levels",
    "oxc_debug_read_u32",
    "oxc_debug_shared_tests_mode",
    "oxc_debug_tri_loads_mode",
    "oxc_debug_set_tuning",
    "oxc_debug_count_occlusion_candidates",
    "oxc_debug_project_aabb",
]


def build(force: bool = False) -> str:
    """Compile liboxcull.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


_libs = {}


def load(path: str = None) -> C.CDLL:
    """liboxcull.so with prototypes.  `path`: another build of the same library (tools/kbench.py compares kernel
    variants built with different -D flags in one process)."""
    path = os.path.abspath(path or os.environ.get("OXC_LIB_PATH") or LIB_PATH)  # OXC_LIB_PATH: an experiment build (tools/build_variants.sh)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the cull path)"
        )
    lib = C.CDLL(path)
    lib.oxc_abi_version.restype = C.c_uint32
    if lib.oxc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.oxc_abi_version()} != {ABI_VERSION} (stale build: run __graft_entry__.build())")
    vp = C.c_void_p
    lib.oxc_abi_version.restype = C.c_uint32
    lib.oxc_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.oxc_destroy.argtypes = [vp]
    lib.oxc_destroy.restype = None
    lib.oxc_last_error.argtypes = [vp]
    lib.oxc_last_error.restype = C.c_char_p
    lib.oxc_reserve.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.oxc_generate_hiz.argtypes = [vp, C.POINTER(MainGeometryContext), vp]
    lib.oxc_cull_geometry.argtypes = [vp, C.POINTER(PreparedFrame), C.POINTER(CullGeometryContext), vp]
    lib.oxc_cull_geometry_batch.argtypes = [vp, C.c_uint32, C.POINTER(PreparedFrame), C.POINTER(CullGeometryContext), vp]
    lib.oxc_join_triangles.argtypes = [vp, vp]
    lib.oxc_seed_meshlet_instances.argtypes = [vp, C.POINTER(CullGeometryContext), C.c_uint32, vp]
    lib.oxc_read_counters.argtypes = [vp, C.POINTER(CullGeometryContext), C.POINTER(Counters), vp]
    lib.oxc_stream_read_probe.argtypes = [vp, vp, C.c_uint64, vp]
    lib.oxc_debug_decode_bounds.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.oxc_profile_begin.argtypes = [vp]
    lib.oxc_profile_end.argtypes = [vp, C.POINTER(KernelTimes)]
    lib.oxc_build_meshlet_bounds.argtypes = [vp, C.POINTER(MeshletBoundsDesc), vp]
    lib.oxc_quantize_vertex_streams.argtypes = [vp, C.POINTER(VertexStreamsDesc), vp]
    lib.oxc_mesh_blob_layout_of.argtypes = [C.POINTER(MeshBlobDesc), C.POINTER(MeshBlobLayout)]
    lib.oxc_mesh_blob_finalize.argtypes = [C.POINTER(MeshBlobDesc), C.POINTER(MeshBlobLayout), C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_float * 6), vp]
    lib.oxc_mesh_build_create.argtypes = [C.POINTER(MeshBuildDesc), C.POINTER(vp)]
    lib.oxc_mesh_build_lod_count.argtypes = [vp]
    lib.oxc_mesh_build_lod_count.restype = C.c_uint32
    lib.oxc_mesh_build_lod.argtypes = [vp, C.c_uint32, C.POINTER(MeshLodView)]
    lib.oxc_mesh_build_destroy.argtypes = [vp]
    lib.oxc_mesh_build_destroy.restype = None
    lib.oxc_mesh_vertex_fetch_remap.argtypes = [vp, C.c_uint64, C.c_uint32, vp, vp]
    lib.oxc_generate_hpb.argtypes = [vp, Buffer, C.POINTER(ImageArrayU8), vp]
    lib.oxc_cull_terrain.argtypes = [vp, C.POINTER(TerrainContext), vp]
    lib.oxc_debug_read_u32.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.oxc_debug_shared_tests_mode.argtypes = [vp]
    lib.oxc_debug_shared_tests_mode.restype = C.c_uint32
    lib.oxc_debug_tri_loads_mode.argtypes = [vp]
    lib.oxc_debug_tri_loads_mode.restype = C.c_uint32
    lib.oxc_debug_raster_stats.argtypes = [vp, vp, vp]
    lib.oxc_debug_set_tuning.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.oxc_debug_count_occlusion_candidates.argtypes = [vp, vp]
    lib.oxc_comm_unique_id.argtypes = [vp, vp]
    lib.oxc_comm_init.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    lib.oxc_comm_destroy.argtypes = [vp]
    lib.oxc_pack_counters.argtypes = [vp, C.POINTER(CullGeometryContext), vp, vp]
    lib.oxc_pack_counters_batch.argtypes = [vp, C.c_uint32, C.POINTER(CullGeometryContext), vp, vp]
    lib.oxc_exchange_counts.argtypes = [vp, vp, vp, vp]
    lib.oxc_broadcast_hiz.argtypes = [vp, C.POINTER(Image), C.c_uint64, C.c_uint32, vp]
    lib.oxc_broadcast_hiz_levels.argtypes = [vp, C.POINTER(Image), C.c_uint32, C.c_uint64, C.c_uint32, vp]
    lib.oxc_debug_project_aabb.argtypes = [vp, C.POINTER(C.c_float), C.c_float, vp, C.c_uint32, vp, vp]
    lib.oxc_draw_visbuffer.argtypes = [vp, C.POINTER(PreparedFrame), C.POINTER(DrawContext), vp]
    for name in EXPORTS:
        if name not in ("oxc_abi_version", "oxc_destroy", "oxc_last_error", "oxc_mesh_build_lod_count", "oxc_mesh_build_destroy"):
            getattr(lib, name).restype = C.c_int
    _libs[path] = lib
    return lib


class MeshBuildDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("vertex_count", C.c_uint32), ("index_count", C.c_uint32), ("max_lods", C.c_uint32), ("max_vertices", C.c_uint32),
                ("max_triangles", C.c_uint32), ("positions", C.c_void_p), ("normals", C.c_void_p), ("indices", C.c_void_p)]


class MeshLodView(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("meshlets", C.c_void_p), ("indirect_vertex_indices", C.c_void_p), ("local_triangle_indices", C.c_void_p),
                ("indices_count", C.c_uint32), ("meshlet_count", C.c_uint32), ("indirect_vertex_indices_count", C.c_uint32),
                ("local_triangle_indices_count", C.c_uint32), ("error", C.c_float), ("_pad", C.c_uint32)]


class KernelTimes(C.Structure):
    _fields_ = [("total_ms", C.c_double * 16), ("launches", C.c_uint32 * 16), ("empty_pair_ms", C.c_double)]


# OXC_K_* (include/oxcull.h); the *_late entries are the LatePass instantiations, timed apart
KERNEL_NAMES = ["prepare_instances", "cull_meshes_scan", "cull_meshes_expand", "cull_meshlets_test", "cull_meshlets_emit",
                "cull_triangles_test", "cull_triangles_emit", "hiz", "cull_meshlets_test_late", "cull_meshlets_emit_late",
                "cull_triangles_test_late", "cull_triangles_emit_late", "draw_visbuffer", "build_meshlet_bounds",
                "multiview_setup", "_15"]


class OxcError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"oxcull status {status}: {msg}")
        self.status = status
