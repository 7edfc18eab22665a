"""HIP path vs the committed golden fixtures -- data, not code: the oracle is not called here."""
import os

import numpy as np
import pytest
import torch

from oxylus_amd.renderer import ImageAttachment, MainGeometryContext

from util import assert_same, gpu_frame, scene_from_golden

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hip_matches_golden_meshlet_stage(renderer):
    s, z = scene_from_golden(os.path.join(GOLDEN, "meshlets_37x111.npz"), "cuda")
    got = gpu_frame(renderer, s, with_triangles=False)
    assert np.array_equal(got["visible"], z["visible"])


def test_hip_matches_golden_pipeline(renderer):
    s, z = scene_from_golden(os.path.join(GOLDEN, "pipeline_12x40.npz"), "cuda")
    hiz = ImageAttachment.hiz(64, 64, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(torch.from_numpy(z["depth"]).cuda()), hiz))
    assert list(hiz.level_offset) == list(z["hiz_offs"])
    assert np.array_equal(hiz.data.cpu().numpy().view(np.uint32), z["hiz"].view(np.uint32))
    got = gpu_frame(renderer, s.clone(), run_cull_meshes=True)
    want = {k[6:]: z[k] for k in z.files if k.startswith("plain_")}
    want["total"], want["cull_meshlets_cmd_x"] = int(want["total"]), int(want["cull_meshlets_cmd_x"])
    assert_same(want, got, ["total", "cull_meshlets_cmd_x", "lod_index", "meshlet_instances", "visible", "indices"])
    got = gpu_frame(renderer, s.clone(), use_hiz=True, hiz=hiz, mask=torch.from_numpy(z["mask_in"]), two_pass=True)
    want = {k[4:]: z[k] for k in z.files if k.startswith("two_")}
    want["early"], want["late"] = int(want["early"]), int(want["late"])
    assert_same(want, got, ["early", "late", "early_visible", "late_visible", "early_indices", "late_indices", "mask"])


def test_hip_matches_golden_widening_rows(renderer):
    """SURVEY 8(f) rows against committed data: bounds producer, HPB producer, terrain cull, draw consumer."""
    from oxylus_amd import lib as L
    from oxylus_amd.renderer import CullGeometryContext, HpbAttachment, PreparedFrame

    z = np.load(os.path.join(GOLDEN, "widening_rows.npz"))
    cu = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    b, m6, q = renderer.build_meshlet_bounds(cu("bounds_positions"), cu("bounds_meshlets"), cu("bounds_vidx"), cu("bounds_micro"))
    assert np.array_equal(b.cpu().numpy(), z["bounds_records"]) and np.array_equal(q.cpu().numpy(), z["bounds_qpos"])
    assert np.array_equal(m6.cpu().numpy().view(np.uint32), z["bounds_mesh6"].view(np.uint32))
    hpb = HpbAttachment.create(9, 6, 3, 4, "cuda")
    renderer.generate_hpb(cu("hpb_page_table"), hpb)
    assert np.array_equal(hpb.data.cpu().numpy(), z["hpb_data"])
    hiz = ImageAttachment.hiz(64, 64, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(cu("terrain_depth")), hiz))
    cam = L.CullCamera()
    for i in range(16):
        cam.projection_view[i] = float(z["terrain_pv"][i])
    cam.near_clip = 0.1
    p = z["terrain_params"]
    mask = torch.from_numpy(z["terrain_mask_in"].copy()).cuda()
    early, _ = renderer.cull_terrain(L.CULL_TEST_ALL, cam, p[0:2], p[2:4], (37, 29), float(p[4]), float(p[5]), cu("terrain_minmax"), mask, hiz=hiz)
    late, cmd = renderer.cull_terrain(L.CULL_TEST_ALL | L.CULL_LATE_PASS, cam, p[0:2], p[2:4], (37, 29), float(p[4]), float(p[5]), cu("terrain_minmax"), mask,
                                      hiz=hiz)
    assert np.array_equal(early.cpu().numpy(), z["terrain_early"]) and np.array_equal(late.cpu().numpy(), z["terrain_late"])
    assert np.array_equal(mask.cpu().numpy(), z["terrain_mask_out"]) and cmd == [4, len(z["terrain_late"]), 0, 0]
    # draw consumer: cull the pipeline fixture on the GPU (== its golden lists, asserted above), then rasterise
    s, zp = scene_from_golden(os.path.join(GOLDEN, "pipeline_12x40.npz"), "cuda")
    frame = PreparedFrame.create(s, expand=False)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_ALL, cull_camera=s.cull_camera())
    renderer.cull_geometry(ctx)
    vd = torch.empty((384, 512), dtype=torch.int64, device="cuda")
    renderer.draw_visbuffer(ctx, zp["camera_pv"], 512, 384, vd, clear=True)
    torch.cuda.synchronize()
    assert np.array_equal(vd.cpu().numpy(), np.load(os.path.join(GOLDEN, "raster_512x384.npz"))["visdepth"])


def test_hip_matches_golden_vertex_streams(renderer):
    """SURVEY 8(f)-1, format side: the three quantised vertex streams against the committed fixture (specials first)."""
    z = np.load(os.path.join(GOLDEN, "vertex_streams_4096.npz"))
    pos, nrm, uv = (torch.from_numpy(z[k]).cuda() for k in ("positions", "normals", "texcoords"))
    qpos, qnrm, quv = renderer.quantize_vertex_streams(pos, nrm, uv)
    torch.cuda.synchronize()
    assert np.array_equal(qpos.cpu().numpy(), z["qpos"])
    assert np.array_equal(qnrm.cpu().numpy(), z["qnrm"])
    assert np.array_equal(quv.cpu().numpy(), z["quv"])
