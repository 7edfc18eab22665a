"""Host-side mirror of the reference's RendererInstance cull entry points over the C ABI.

Same names and field meaning as Oxylus/include/Render/RendererInstance.hpp:143-216,397-398:
`RendererInstance.generate_hiz(MainGeometryContext)` and
`RendererInstance.cull_geometry(CullGeometryContext)`.  Buffers are torch CUDA tensors (device
memory + streams only; all compute is in liboxcull.so).  The compiled C++ shim with the same
surface is oxylus_amd/host/RendererInstance.hpp; this Python twin exists because the test
runner and bench driver are Python.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import lib as L
from .synth import Scene, hiz_layout


def _buf(t: Optional[torch.Tensor]) -> L.Buffer:
    if t is None:
        return L.Buffer(None, 0)
    return L.Buffer(t.data_ptr(), t.numel() * t.element_size())


@dataclass
class ImageAttachment:
    """Linear R32F mip chain (stand-in for vuk::ImageAttachment)."""
    data: torch.Tensor  # float32 1-D
    width: int
    height: int
    levels: int
    level_offset: list  # bytes

    @staticmethod
    def hiz(width: int, height: int, device, levels: Optional[int] = None) -> "ImageAttachment":
        levels, offs, total = hiz_layout(width, height, levels)
        # vuk::clear_image(hiz, DepthZero), RendererInstance.cpp:588
        return ImageAttachment(torch.zeros(total // 4, dtype=torch.float32, device=device), width, height, levels, offs)

    @staticmethod
    def depth(t: torch.Tensor) -> "ImageAttachment":
        assert t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
        return ImageAttachment(t.view(-1), t.shape[1], t.shape[0], 1, [0])

    def level(self, k: int) -> torch.Tensor:
        w, h = max(1, self.width >> k), max(1, self.height >> k)
        o = self.level_offset[k] // 4
        return self.data[o:o + w * h].view(h, w)

    def c(self) -> L.Image:
        im = L.Image()
        im.dptr = self.data.data_ptr()
        im.width, im.height, im.levels = self.width, self.height, self.levels
        for k, o in enumerate(self.level_offset):
            im.level_offset[k] = o
        return im


@dataclass
class HpbAttachment:
    """R8UI Texture2DArray with mips (VSM hierarchical page buffer), linear layout."""
    data: torch.Tensor  # uint8 1-D
    width: int
    height: int
    layers: int
    levels: int
    level_offset: list  # bytes

    @staticmethod
    def create(width: int, height: int, layers: int, levels: int, device) -> "HpbAttachment":
        offs, off = [], 0
        for k in range(levels):
            offs.append(off)
            off += layers * max(1, width >> k) * max(1, height >> k)
            off = (off + 255) // 256 * 256
        return HpbAttachment(torch.zeros(off, dtype=torch.uint8, device=device), width, height, layers, levels, offs)

    def level(self, k: int) -> torch.Tensor:
        w, h = max(1, self.width >> k), max(1, self.height >> k)
        o = self.level_offset[k]
        return self.data[o:o + self.layers * w * h].view(self.layers, h, w)

    def build_mips(self):
        """Any-bit pyramid from level 0 (what rmvsm_downsample_hpb produces: a texel is set if
        any of its 2x2 children is)."""
        for k in range(1, self.levels):
            p = self.level(k - 1)
            w, h = max(1, self.width >> k), max(1, self.height >> k)
            self.level(k).copy_(p.view(self.layers, h, 2, w, 2).amax(dim=(2, 4)) if p.shape[1] >= 2 and p.shape[2] >= 2 else p.amax(dim=(1, 2), keepdim=True))

    def c(self) -> L.ImageArrayU8:
        im = L.ImageArrayU8()
        im.dptr = self.data.data_ptr()
        im.width, im.height, im.layers, im.levels = self.width, self.height, self.layers, self.levels
        for k, o in enumerate(self.level_offset):
            im.level_offset[k] = o
        return im


@dataclass
class PreparedFrame:
    """The PreparedFrame buffers of the cull path (RendererInstance.hpp:143-169), sized as in
    RendererInstance::update (RendererInstance.cpp:1640-1732)."""
    scene: Scene
    max_meshlet_instance_count: int
    meshlet_instances_buffer: torch.Tensor
    visible_meshlet_instances_indices_buffer: torch.Tensor
    meshlet_instance_visibility_mask_buffer: torch.Tensor
    reordered_indices_buffer: Optional[torch.Tensor]

    @staticmethod
    def create(scene: Scene, with_triangles: bool = True, expand: bool = True, max_tris: int = 64, index_words: int = 1) -> "PreparedFrame":
        """max_tris = 128 for wide_triangle_index != 0; index_words = 2 for wide_triangle_index = 2 ({id, corner} pairs, 8 bytes per index)."""
        dev = scene.device
        n = scene.n_meshlet_instances
        mli = scene.meshlet_instances.clone() if expand else torch.zeros((n, 2), dtype=torch.int32, device=dev)
        vis_idx = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        # zero-filled on (re)upload of the instances, RendererInstance.cpp:1651-1665
        mask = torch.zeros(max((n + 31) // 32, 1), dtype=torch.int32, device=dev)
        reordered = torch.zeros(max(n, 1) * max_tris * 3 * index_words, dtype=torch.int32, device=dev) if with_triangles else None
        return PreparedFrame(scene, n, mli, vis_idx, mask, reordered)

    def c(self) -> L.PreparedFrame:
        f = L.PreparedFrame()
        s = self.scene
        f.mesh_instance_count = s.n_mesh_instances
        f.max_meshlet_instance_count = self.max_meshlet_instance_count
        f.meshes_buffer = _buf(s.meshes)
        f.transforms_world_buffer = _buf(s.transforms)
        f.mesh_instances_buffer = _buf(s.mesh_instances)
        f.meshlet_instances_buffer = _buf(self.meshlet_instances_buffer)
        f.visible_meshlet_instances_indices_buffer = _buf(self.visible_meshlet_instances_indices_buffer)
        f.meshlet_instance_visibility_mask_buffer = _buf(self.meshlet_instance_visibility_mask_buffer)
        f.reordered_indices_buffer = _buf(self.reordered_indices_buffer)
        return f


@dataclass
class CullGeometryContext:
    """RendererInstance.hpp:171-197."""
    use_hiz: bool = False
    use_hpb: bool = False
    init_cull_meshes: bool = False
    cull_flags: int = L.CULL_TEST_ALL
    cull_camera: Optional[L.CullCamera] = None
    hiz_attachment: Optional[ImageAttachment] = None
    hpb_attachment: Optional[HpbAttachment] = None
    vsm_clipmaps_buffer: Optional[torch.Tensor] = None            # uint8 [V*76] GPU::VirtualClipmap records
    vsm_clipmap_dirty_flags_buffer: Optional[torch.Tensor] = None  # int32 [V]
    vsm_clipmap_count: int = 0
    wide_triangle_index: int = 0  # extension: 1 / True = (id << 9) | (3t+k), meshlets of up to 128 triangles, <= 2^23 ids; 2 = {id, corner} pairs, no id limit
    small_triangle_cull: bool = False  # extension (north star): also drop triangles whose screen bbox covers no pixel centre
    async_triangles: bool = False  # extension (scheduling only): the triangle stage runs on the context's own stream; RendererInstance.join_triangles
    share_pass_tests: bool = False  # extension (caching only): the late HiZ call of a frame reuses the early call's frustum + cone results (include/oxcull.h)
    unordered_output: int = 0  # extension (order only): 0 ascending lists, 1 the reference's atomic slot allocation where it is faster (include/oxcull.h)
    implicit_meshlet_instances: bool = False  # extension (multi-view batch only): the MeshletInstance list stays implicit, runs in meshlet_instance_runs_buffer
    meshlet_instance_runs_buffer: Optional[torch.Tensor] = None  # int32 [M, 2] {first, count} per mesh instance (out)
    stages: int = 0
    _c: L.CullGeometryContext = field(default_factory=L.CullGeometryContext)

    def c(self) -> L.CullGeometryContext:
        c = self._c
        c.struct_size = C.sizeof(L.CullGeometryContext)
        c.use_hiz, c.use_hpb, c.init_cull_meshes = int(self.use_hiz), int(self.use_hpb), int(self.init_cull_meshes)
        c.cull_flags, c.stages = self.cull_flags, self.stages
        if self.cull_camera is not None:
            c.cull_camera = self.cull_camera
        if self.hiz_attachment is not None:
            c.hiz_attachment = self.hiz_attachment.c()
        if self.hpb_attachment is not None:
            c.hpb_attachment = self.hpb_attachment.c()
        c.vsm_clipmaps_buffer = _buf(self.vsm_clipmaps_buffer)
        c.vsm_clipmap_dirty_flags_buffer = _buf(self.vsm_clipmap_dirty_flags_buffer)
        c.vsm_clipmap_count = self.vsm_clipmap_count
        c.wide_triangle_index = int(self.wide_triangle_index)
        c.small_triangle_cull = int(self.small_triangle_cull)
        c.async_triangles = int(self.async_triangles)
        c.share_pass_tests = int(self.share_pass_tests)
        c.unordered_output = int(self.unordered_output)
        c.implicit_meshlet_instances = int(self.implicit_meshlet_instances)
        c.meshlet_instance_runs_buffer = _buf(self.meshlet_instance_runs_buffer)
        return c


@dataclass
class MainGeometryContext:
    """The fields generate_hiz uses (RendererInstance.hpp:199-216)."""
    depth_attachment: ImageAttachment
    hiz_attachment: ImageAttachment


class RendererInstance:
    """Owns one oxc_ctx on one device."""

    def __init__(self, device_index: int = 0, lib_path: str = None):
        self._lib = L.load(lib_path)
        if not torch.cuda.is_available():
            raise RuntimeError("oxylus_amd.RendererInstance needs a GPU: the cull path has no CPU fallback")
        self.device_index = device_index
        self._ctx = C.c_void_p()
        st = self._lib.oxc_create(device_index, C.byref(self._ctx))
        if st != L.OXC_OK:
            raise L.OxcError(st, "oxc_create failed")
        self.prepared_frame: Optional[PreparedFrame] = None

    def close(self):
        if self._ctx:
            self._lib.oxc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != L.OXC_OK:
            raise L.OxcError(st, self._lib.oxc_last_error(self._ctx).decode())

    @staticmethod
    def _stream(stream) -> C.c_void_p:
        s = stream if stream is not None else torch.cuda.current_stream()
        return C.c_void_p(s.cuda_stream)

    def reserve(self, max_mesh_instances: int, max_meshlet_instances: int):
        self._check(self._lib.oxc_reserve(self._ctx, max_mesh_instances, max_meshlet_instances))

    def generate_hiz(self, context: MainGeometryContext, stream=None):
        c = L.MainGeometryContext()
        c.struct_size = C.sizeof(L.MainGeometryContext)
        c.depth_attachment = context.depth_attachment.c()
        c.hiz_attachment = context.hiz_attachment.c()
        self._check(self._lib.oxc_generate_hiz(self._ctx, C.byref(c), self._stream(stream)))

    def cull_geometry(self, context: CullGeometryContext, stream=None):
        assert self.prepared_frame is not None
        f = self.prepared_frame.c()
        self._check(self._lib.oxc_cull_geometry(self._ctx, C.byref(f), C.byref(context.c()), self._stream(stream)))

    def join_triangles(self, stream=None):
        """`stream` waits for every triangle stage still in flight on the context's own stream (async_triangles)."""
        self._check(self._lib.oxc_join_triangles(self._ctx, self._stream(stream)))

    def cull_geometry_batch(self, frames, contexts, stream=None):
        """Batched cull of independent (PreparedFrame, CullGeometryContext) pairs (oxc_cull_geometry_batch).
        The contexts' output buffers are updated like cull_geometry does."""
        n = len(frames)
        assert n == len(contexts) and n > 0
        cf = (L.PreparedFrame * n)(*[f.c() for f in frames])
        cc = (L.CullGeometryContext * n)(*[c.c() for c in contexts])
        self._check(self._lib.oxc_cull_geometry_batch(self._ctx, n, cf, cc, self._stream(stream)))
        for i, c in enumerate(contexts):
            for name in ("visibility_buffer", "cull_meshlets_cmd_buffer", "cull_triangles_cmd_buffer", "draw_geometry_cmd_buffer"):
                setattr(c._c, name, getattr(cc[i], name))

    def seed_meshlet_instances(self, context: CullGeometryContext, total: int, stream=None):
        self._check(self._lib.oxc_seed_meshlet_instances(self._ctx, C.byref(context.c()), total, self._stream(stream)))

    def read_counters(self, context: CullGeometryContext, stream=None) -> L.Counters:
        out = L.Counters()
        self._check(self._lib.oxc_read_counters(self._ctx, C.byref(context._c), C.byref(out), self._stream(stream)))
        return out

    def stream_read_probe(self, t: torch.Tensor, stream=None):
        self._check(self._lib.oxc_stream_read_probe(self._ctx, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), self._stream(stream)))

    def debug_decode_bounds(self, bounds: torch.Tensor) -> torch.Tensor:
        n = bounds.shape[0]
        out = torch.empty((n, 10), dtype=torch.float32, device=bounds.device)
        self._check(self._lib.oxc_debug_decode_bounds(self._ctx, C.c_void_p(bounds.data_ptr()), n, C.c_void_p(out.data_ptr()), self._stream(None)))
        return out

    def build_meshlet_bounds(self, positions: torch.Tensor, meshlets: torch.Tensor, vidx: torch.Tensor, micro: torch.Tensor,
                             quantize_positions: bool = True, stream=None):
        """SURVEY 8(f)-1, AssetManager_GLTF.cpp:573-578,683-744: positions f32 [V,3], meshlets i32 [M,4] (GPU::Meshlet),
        vidx i32, micro u8 -> (MeshletBounds as i16 [M,8], mesh bounds f32 [6] = center xyz + extent xyz, u16x4 positions as i16 [V,4])."""
        dev = positions.device
        V, M = positions.shape[0], meshlets.shape[0]
        bounds = torch.empty((M, 8), dtype=torch.int16, device=dev)
        mesh6 = torch.empty(6, dtype=torch.float32, device=dev)
        qpos = torch.empty((V, 4), dtype=torch.int16, device=dev) if quantize_positions else None

        def buf(t):
            return L.Buffer(C.c_void_p(t.data_ptr()), t.numel() * t.element_size()) if t is not None and t.numel() else L.Buffer(None, 0)

        d = L.MeshletBoundsDesc()
        d.struct_size = C.sizeof(L.MeshletBoundsDesc)
        d.vertex_count, d.meshlet_count = V, M
        d.positions, d.meshlets = buf(positions), buf(meshlets)
        d.indirect_vertex_indices, d.local_triangle_indices = buf(vidx), buf(micro)
        d.meshlet_bounds, d.mesh_bounds, d.quantized_positions = buf(bounds), buf(mesh6), buf(qpos)
        self._keep = (positions, meshlets, vidx, micro)
        self._check(self._lib.oxc_build_meshlet_bounds(self._ctx, C.byref(d), self._stream(stream)))
        return bounds, mesh6, qpos

    def quantize_vertex_streams(self, positions: torch.Tensor = None, normals: torch.Tensor = None, texcoords: torch.Tensor = None, stream=None):
        """AssetManager_GLTF.cpp:570-588: f32 [V,3] positions -> i16 [V,4] (u16x4 halfs), f32 [V,3] normals -> i32 [V] (10:10:10),
        f32 [V,2] texcoords -> i16 [V,2]; absent streams come back as None."""
        given = [t for t in (positions, normals, texcoords) if t is not None]
        if not given:
            return None, None, None
        dev, V = given[0].device, given[0].shape[0]
        qpos = torch.empty((V, 4), dtype=torch.int16, device=dev) if positions is not None else None
        qnrm = torch.empty(V, dtype=torch.int32, device=dev) if normals is not None else None
        quv = torch.empty((V, 2), dtype=torch.int16, device=dev) if texcoords is not None else None

        def buf(t):
            return L.Buffer(C.c_void_p(t.data_ptr()), t.numel() * t.element_size()) if t is not None and t.numel() else L.Buffer(None, 0)

        d = L.VertexStreamsDesc()
        d.struct_size, d.vertex_count = C.sizeof(L.VertexStreamsDesc), V
        d.positions, d.normals, d.texcoords = buf(positions), buf(normals), buf(texcoords)
        d.quantized_positions, d.quantized_normals, d.quantized_texcoords = buf(qpos), buf(qnrm), buf(quv)
        self._keep = (positions, normals, texcoords)
        self._check(self._lib.oxc_quantize_vertex_streams(self._ctx, C.byref(d), self._stream(stream)))
        return qpos, qnrm, quv

    def generate_hpb(self, page_table: torch.Tensor, hpb: "HpbAttachment", stream=None):
        """SURVEY 8(f)-3, Shadowmaps.cpp:331-366: page_table int32 [layers, h, w] -> every level of `hpb`."""
        im = hpb.c()
        self._keep = (page_table, hpb)
        self._check(self._lib.oxc_generate_hpb(self._ctx, L.Buffer(C.c_void_p(page_table.data_ptr()), page_table.numel() * 4), C.byref(im), self._stream(stream)))

    def cull_terrain(self, cull_flags: int, cull_camera, world_min, world_size, patch_count, base_height: float, height_scale: float,
                     patch_minmax: torch.Tensor, mask: torch.Tensor, hiz: "ImageAttachment" = None, stream=None):
        """SURVEY 8(f)-4, Terrain.cpp:159-216 + terrain_cull.slang: patch_minmax f32 [py, px, 2], mask int32 [ceil(total/32)] (in/out).
        Returns (visible_patches int32 [count], count)."""
        pcx, pcy = int(patch_count[0]), int(patch_count[1])
        total = pcx * pcy
        visible = torch.full((max(total, 1),), -1, dtype=torch.int32, device=patch_minmax.device)
        c = L.TerrainContext()
        c.struct_size = C.sizeof(L.TerrainContext)
        c.cull_flags = cull_flags
        c.cull_camera = cull_camera
        c.world_min[0], c.world_min[1] = float(world_min[0]), float(world_min[1])
        c.world_size[0], c.world_size[1] = float(world_size[0]), float(world_size[1])
        c.patch_count[0], c.patch_count[1] = pcx, pcy
        c.base_height, c.height_scale = float(base_height), float(height_scale)
        mm = L.Image()
        mm.dptr, mm.width, mm.height, mm.levels = patch_minmax.data_ptr(), pcx, pcy, 1
        c.patch_minmax_attachment = mm
        if hiz is not None:
            c.hiz_attachment = hiz.c()
        c.visible_patches_buffer = L.Buffer(C.c_void_p(visible.data_ptr()), visible.numel() * 4)
        c.patch_visibility_mask_buffer = L.Buffer(C.c_void_p(mask.data_ptr()), mask.numel() * 4)
        self._keep = (patch_minmax, mask, visible, hiz)
        self._check(self._lib.oxc_cull_terrain(self._ctx, C.byref(c), self._stream(stream)))
        host = (C.c_uint32 * 4)()
        self._check(self._lib.oxc_debug_read_u32(self._ctx, c.draw_cmd_buffer.dptr, 4, C.cast(host, C.c_void_p), self._stream(stream)))
        cmd_host = list(host)
        return visible[: cmd_host[1]].clone(), cmd_host

    def draw_visbuffer(self, context: CullGeometryContext, projection_view, width: int, height: int, visdepth: torch.Tensor, clear: bool,
                       depth: "ImageAttachment" = None, visbuffer: torch.Tensor = None, stream=None, draw_cmd: torch.Tensor = None):
        """SURVEY 8(f)-2, DrawGeometry.cpp:104-190: rasterise the triangles the last cull_geometry(context) emitted into
        `visdepth` (int64 [h, w]: depth bits << 32 | vis); optional resolves into `depth` (ImageAttachment, levels = 1) and
        `visbuffer` (int32 [h, w]).  `draw_cmd` (int32 [5], VkDrawIndexedIndirectCommand) replaces the context's own command:
        draw a caller-written `reordered_indices_buffer`."""
        assert self.prepared_frame is not None
        f = self.prepared_frame.c()
        d = L.DrawContext()
        d.struct_size = C.sizeof(L.DrawContext)
        d.wide_triangle_index = int(context.wide_triangle_index)
        d.clear = 1 if clear else 0
        d.width, d.height = width, height
        for i in range(16):
            d.projection_view[i] = float(projection_view[i])
        d.draw_geometry_cmd_buffer = context._c.draw_geometry_cmd_buffer if draw_cmd is None else L.Buffer(C.c_void_p(draw_cmd.data_ptr()), draw_cmd.numel() * 4)
        d.visdepth_buffer = L.Buffer(C.c_void_p(visdepth.data_ptr()), visdepth.numel() * 8)
        if depth is not None:
            d.depth_attachment = depth.c()
        if visbuffer is not None:
            d.visbuffer_attachment = L.Buffer(C.c_void_p(visbuffer.data_ptr()), visbuffer.numel() * 4)
        self._keep = (visdepth, depth, visbuffer, draw_cmd)
        self._check(self._lib.oxc_draw_visbuffer(self._ctx, C.byref(f), C.byref(d), self._stream(stream)))

    def debug_raster_stats(self, stream=None) -> dict:
        """What the last draw_visbuffer did with its triangles (test hook; synchronises)."""
        out = (C.c_uint32 * 4)()
        self._check(self._lib.oxc_debug_raster_stats(self._ctx, C.cast(out, C.c_void_p), self._stream(stream)))
        return {"big": int(out[0]), "clipped": int(out[1]), "tiles": int(out[2]), "overflowed_segments": int(out[3])}

    def debug_set_tuning(self, knob: int, value: int):
        """Harness knobs (L.TUNE_*): async stage grid caps, the raster queues' capacity (before the first draw)."""
        self._check(self._lib.oxc_debug_set_tuning(self._ctx, knob, value))

    def debug_count_occlusion_candidates(self, counters):
        """Measurement aid (include/oxcull.h): `counters` = int32 CUDA tensor of 256 * 64 zeros (or None to switch it off); the HiZ calls that follow
        run the counting instantiations of their meshlet test, which add the candidates that reach test_occlusion to it (sum = the count)."""
        self._check(self._lib.oxc_debug_count_occlusion_candidates(self._ctx, C.c_void_p(counters.data_ptr()) if counters is not None else None))

    def debug_shared_tests_mode(self) -> int:
        """What share_pass_tests did in the last cull_geometry call: 0 tested on its own, 1 early call that published, 2 late call that reused,
        3 late call that reused and needed no prepare kernel."""
        return int(self._lib.oxc_debug_shared_tests_mode(self._ctx))

    def debug_tri_loads_mode(self) -> int:
        """1 = the last call's triangle kernels used nt loads, 2 = plain loads (shared geometry), 0 = no triangle stage ran."""
        return int(self._lib.oxc_debug_tri_loads_mode(self._ctx))

    def debug_project_aabb(self, mvp16, near_clip: float, boxes6: torch.Tensor) -> torch.Tensor:
        """boxes6 f32 [n, 6] = {center.xyz, extent.xyz} -> f32 [n, 7] = {min.u, min.v, min.z, max.u, max.v, max.z, valid}."""
        n = boxes6.shape[0]
        out = torch.empty((n, 7), dtype=torch.float32, device=boxes6.device)
        m = (C.c_float * 16)(*[float(v) for v in mvp16])
        self._check(self._lib.oxc_debug_project_aabb(self._ctx, m, float(near_clip), C.c_void_p(boxes6.data_ptr()), n, C.c_void_p(out.data_ptr()), self._stream(None)))
        return out

    # ---- multi-GPU exchange through the C ABI (RCCL): SURVEY 8e ----
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self._lib.oxc_comm_unique_id(self._ctx, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._check(self._lib.oxc_comm_init(self._ctx, C.c_char_p(unique_id), rank, world))
        self._comm_world = world

    def comm_destroy(self):
        self._check(self._lib.oxc_comm_destroy(self._ctx))

    def pack_counters(self, context: CullGeometryContext, counts4: torch.Tensor, stream=None):
        """{emitted, early, late, index_count} of the context's last call -> counts4 (int32 [4], device), on the stream."""
        self._check(self._lib.oxc_pack_counters(self._ctx, C.byref(context._c), C.c_void_p(counts4.data_ptr()), self._stream(stream)))

    def exchange_counts(self, counts4: torch.Tensor, stream=None) -> torch.Tensor:
        """counts4: int32 [4] on the device -> int32 [world, 4] (all-gather on the stream)."""
        out = torch.empty((self._comm_world, 4), dtype=torch.int32, device=counts4.device)
        self._keep = (counts4, out)
        self._check(self._lib.oxc_exchange_counts(self._ctx, C.c_void_p(counts4.data_ptr()), C.c_void_p(out.data_ptr()), self._stream(stream)))
        return out

    def broadcast_hiz(self, hiz: "ImageAttachment", root: int, stream=None, first_level: int = 0):
        """Every level (first_level = 0) or only the top of the pyramid, levels >= first_level (the rest is built locally)."""
        im = hiz.c()
        if first_level:
            self._check(self._lib.oxc_broadcast_hiz_levels(self._ctx, C.byref(im), first_level, hiz.data.numel() * 4, root, self._stream(stream)))
        else:
            self._check(self._lib.oxc_broadcast_hiz(self._ctx, C.byref(im), hiz.data.numel() * 4, root, self._stream(stream)))

    def profile_begin(self):
        self._check(self._lib.oxc_profile_begin(self._ctx))

    def profile_end(self) -> dict:
        kt = L.KernelTimes()
        self._check(self._lib.oxc_profile_end(self._ctx, C.byref(kt)))
        out = {"empty_pair_ms": kt.empty_pair_ms, "kernels": {}}
        for i, name in enumerate(L.KERNEL_NAMES):
            if kt.launches[i]:
                out["kernels"][name] = {"launches": int(kt.launches[i]), "total_ms": float(kt.total_ms[i])}
        return out
