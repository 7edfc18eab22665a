"""Shared helpers of the parity tests: run one configuration through the CPU oracle and through
the HIP path (via the C ABI) on the same seeded scene and return comparable results."""
from __future__ import annotations

import numpy as np
import torch

import oracle
from oxylus_amd import lib as L
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame
from oxylus_amd.synth import hiz_layout


def oracle_hiz(depth_cpu: torch.Tensor, w: int, h: int, levels=None):
    levels, offs, total = hiz_layout(w, h, levels)
    data = torch.zeros(total // 4, dtype=torch.float32)
    oracle.generate_hiz(depth_cpu, data, w, h, levels, offs)
    return data, levels, offs


def oracle_frame(scene_cpu, cull_flags=L.CULL_TEST_ALL, use_hiz=False, hiz=None, mask=None, run_cull_meshes=False,
                 two_pass=False, with_triangles=True):
    """Runs the reference sequence on the CPU oracle.  Returns a dict of numpy arrays."""
    cam = scene_cpu.cull_camera()
    res = {}
    if run_cull_meshes:
        mli, cmd = oracle.cull_meshes(scene_cpu, cam, cull_flags)
        res["cull_meshlets_cmd_x"] = int(cmd[0])
        res["lod_index"] = scene_cpu.mesh_instances[:, 1].numpy().copy()
    else:
        mli = scene_cpu.meshlet_instances
    res["meshlet_instances"] = mli.numpy().copy()
    n = mli.shape[0]
    res["total"] = n
    if not use_hiz:
        vis = oracle.cull_meshlets(scene_cpu, cam, mli)
        res["visible"] = vis.numpy().copy()
        if with_triangles:
            res["indices"] = oracle.cull_triangles(scene_cpu, cam, mli, vis, 0, vis.numel()).numpy().copy()
        return res
    hz = oracle.make_hiz(hiz["data"], hiz["w"], hiz["h"], hiz["levels"], hiz["offs"])
    v = oracle.Visibility(n, 0, 0)
    out = torch.zeros(max(n, 1), dtype=torch.int32)
    mask = mask.clone()
    passes = [cull_flags, cull_flags | L.CULL_LATE_PASS] if two_pass else [cull_flags]
    for i, flags in enumerate(passes):
        tag = "late" if (flags & L.CULL_LATE_PASS) else "early"
        emitted = oracle.cull_meshlets_hiz(scene_cpu, cam, mli, flags, hz, v, mask, out)
        first = v.early if (flags & L.CULL_LATE_PASS) else 0
        res[f"{tag}_emitted"] = emitted
        res[f"{tag}_visible"] = out[first:first + emitted].numpy().copy()
        if with_triangles:
            res[f"{tag}_indices"] = oracle.cull_triangles(scene_cpu, cam, mli, out, first, emitted).numpy().copy()
    res["early"], res["late"] = v.early, v.late
    res["mask"] = mask.numpy().copy()
    return res


def gpu_frame(renderer, scene_gpu, cull_flags=L.CULL_TEST_ALL, use_hiz=False, hiz: ImageAttachment = None, mask=None,
              run_cull_meshes=False, two_pass=False, with_triangles=True, share_pass_tests=False, before_pass=None, unordered_output=0,
              wide_triangle_index=False, small_triangle_cull=False, max_tris=64):
    """Same sequence through liboxcull.so.  Returns numpy arrays in the layout of oracle_frame.
    share_pass_tests: the flag of include/oxcull.h on every call; before_pass(i, ctx): called in front of call i (tests that change
    something between the early and the late call)."""
    words = 2 if int(wide_triangle_index) == 2 else 1  # {id, corner} pairs: two words per index
    frame = PreparedFrame.create(scene_gpu, with_triangles=with_triangles, expand=not run_cull_meshes, max_tris=max_tris, index_words=words)
    if mask is not None:
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask.to(scene_gpu.device))
    renderer.prepared_frame = frame
    cam = scene_gpu.cull_camera()
    stages = L.STAGE_ALL if with_triangles else (L.STAGE_MESHES | L.STAGE_MESHLETS)
    ctx = CullGeometryContext(use_hiz=use_hiz, init_cull_meshes=run_cull_meshes, cull_flags=cull_flags, cull_camera=cam,
                              hiz_attachment=hiz, stages=stages, share_pass_tests=share_pass_tests, unordered_output=unordered_output,
                              wide_triangle_index=wide_triangle_index, small_triangle_cull=small_triangle_cull)
    res = {}
    if not run_cull_meshes:
        renderer.seed_meshlet_instances(ctx, scene_gpu.n_meshlet_instances)
    passes = [cull_flags, cull_flags | L.CULL_LATE_PASS] if (use_hiz and two_pass) else [cull_flags]
    for i, flags in enumerate(passes):
        ctx.cull_flags = flags
        if i > 0:
            ctx.init_cull_meshes = False
        if before_pass is not None:
            before_pass(i, ctx)
        renderer.cull_geometry(ctx)
        res.setdefault("share_modes", []).append(renderer.debug_shared_tests_mode())
        c = renderer.read_counters(ctx)
        late = bool(flags & L.CULL_LATE_PASS)
        tag = ("late" if late else "early") if use_hiz else None
        first = c.early_visible_meshlet_instances if (use_hiz and late) else 0
        emitted = c.cull_triangles_cmd_x
        vis = frame.visible_meshlet_instances_indices_buffer[first:first + emitted].cpu().numpy().copy()
        idx = frame.reordered_indices_buffer[:c.draw_index_count * words].cpu().numpy().copy() if with_triangles else None
        if use_hiz:
            res[f"{tag}_emitted"] = emitted
            res[f"{tag}_visible"] = vis
            if with_triangles:
                res[f"{tag}_indices"] = idx
        else:
            res["visible"] = vis
            if with_triangles:
                res["indices"] = idx
        res["total"] = c.total_visible_meshlet_instances
        res["cull_meshlets_cmd_x"] = c.cull_meshlets_cmd_x
        res["early"], res["late"] = c.early_visible_meshlet_instances, c.late_visible_meshlet_instances
    res["meshlet_instances"] = frame.meshlet_instances_buffer[:res["total"]].cpu().numpy().copy()
    res["lod_index"] = scene_gpu.mesh_instances[:, 1].cpu().numpy().copy()
    res["mask"] = frame.meshlet_instance_visibility_mask_buffer.cpu().numpy().copy()
    return res


def sorted_lists(res: dict) -> dict:
    """unordered_output: the lists of a frame as ascending u32 arrays -- what the ordered form emits (packed indices ascend with
    (meshlet instance, triangle, corner), so sorting the whole index list restores the ordered bytes)."""
    out = dict(res)
    for k, v in res.items():
        if isinstance(v, np.ndarray) and (k.endswith("visible") or k.endswith("indices")):
            out[k] = np.sort(v.view(np.uint32)).view(v.dtype)
    return out


def pairs_as_u64(indices: np.ndarray) -> np.ndarray:
    """wide_triangle_index = 2: the {u32 id, u32 corner} pairs of an index list as (id << 32) | corner -- ascending exactly when the ordered list is."""
    u = indices.view(np.uint32).reshape(-1, 2).astype(np.uint64)
    return (u[:, 0] << np.uint64(32)) | u[:, 1]


def assert_triangles_adjacent(indices: np.ndarray, corner_bits: int = 8):
    """A packed index list is a sequence of triangles: (id << bits) | 3t, | 3t + 1, | 3t + 2 next to each other, whatever the order."""
    u = indices.view(np.uint32).reshape(-1, 3)
    assert np.all(u[:, 1] == u[:, 0] + 1) and np.all(u[:, 2] == u[:, 0] + 2), "a triangle's three packed indices are not adjacent"
    assert np.all((u[:, 0] & ((1 << corner_bits) - 1)) % 3 == 0)


def assert_same(a: dict, b: dict, keys):
    for k in keys:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray) or isinstance(y, np.ndarray):
            x, y = np.asarray(x), np.asarray(y)
            assert x.shape == y.shape, f"{k}: shape {x.shape} vs {y.shape}"
            assert np.array_equal(x, y), f"{k}: {int((x != y).sum())} of {x.size} elements differ"
        else:
            assert x == y, f"{k}: {x} vs {y}"


def scene_from_golden(path: str, device="cpu"):
    """Rebuild a Scene (reference layouts, pointers re-bound for `device`) from a golden .npz."""
    from oxylus_amd.synth import Scene, SceneSpec

    z = np.load(path)
    M, K, Lc, n_meshes = [int(x) for x in z["spec"]]
    dev = torch.device(device)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    lods = torch.zeros((z["lods32"].shape[0], 8), dtype=torch.int64)
    lods.view(torch.int32)[:, 10:16] = torch.from_numpy(z["lods32"])
    meshes = torch.zeros((n_meshes, 8), dtype=torch.int64)
    m32 = meshes.view(torch.int32)
    m32[:, 6:8] = torch.from_numpy(z["meshes32"])
    m32[:, 10:16] = torch.from_numpy(z["mesh_bounds"])
    misc = z["camera_misc"]
    camera = {"projection_view": [float(x) for x in z["camera_pv"]], "position": [float(x) for x in misc[0:3]],
              "acceptable_lod_error": float(misc[3]), "resolution": [float(misc[4]), float(misc[5])], "near_clip": float(misc[6])}
    tables = {k: torch.from_numpy(z[k]) for k in ("meshlet_start", "vidx_start", "micro_start", "mesh_vertex_start")}
    s = Scene(spec=SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, lod_count=Lc), device=dev, bounds=T(z["bounds"]), meshlets=T(z["meshlets"]),
              micro=T(z["micro"]), vidx=T(z["vidx"]), positions=T(z["positions"]), lods=lods.to(dev), meshes=meshes.to(dev),
              transforms=T(z["transforms"]), mesh_instances=T(z["mesh_instances"]), meshlet_instances=T(z["meshlet_instances"]),
              camera=camera, n_meshes=n_meshes, lod_meshlet_counts=None, _lod_tables=tables)
    return s.bind(), z
