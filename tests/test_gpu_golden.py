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
