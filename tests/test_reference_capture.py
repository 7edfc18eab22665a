"""The way the oracle gets PINNED.

tests/golden/reference_capture_*.npz -- buffers dumped from a Vulkan run of the reference on a scene exported by
tools/reference_capture/export_scene.py (format: tools/reference_capture/README.md) -- are compared with the checker (and, on the GPU box,
with the HIP path) by tests/capture_compare.py: counts equal, meshlet lists equal as sorted sets, mask and pyramid byte-identical,
triangle lists equal modulo the checker's boundary set.  No capture exists yet (no Vulkan device in the authoring container): those
tests SKIP, and the comparator is exercised by captures synthesised from the fast-math-envelope build of the checker (a legal
re-association of the same shaders: its flips must all fall inside the boundary set) and by corrupted ones that must be refused."""
import glob
import os

import numpy as np
import pytest
import torch

from capture_compare import CAPTURE_VERSION, compare_capture
from util import oracle_frame, oracle_hiz, scene_from_golden

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CAPTURES = sorted(glob.glob(os.path.join(GOLDEN, "reference_capture_*.npz")))


def _expected(scene_file: str):
    s, z = scene_from_golden(os.path.join(GOLDEN, scene_file))
    hz, levels, offs = oracle_hiz(torch.from_numpy(z["depth"]), 64, 64)
    hizd = {"data": hz, "w": 64, "h": 64, "levels": levels, "offs": offs}
    two = oracle_frame(s.clone(), use_hiz=True, hiz=hizd, mask=torch.from_numpy(z["mask_in"]), two_pass=True)
    return s, z, hizd, two


def _synth_capture(oracle_lib, scene_file: str, seed: int = 1) -> dict:
    """What a Vulkan run could have produced: the fast-math envelope's decisions, every list in a scrambled (atomic) order."""
    s, z, hizd, _ = _expected(scene_file)
    with oracle_lib.variant("fast"):
        two = oracle_frame(s.clone(), use_hiz=True, hiz=hizd, mask=torch.from_numpy(z["mask_in"]), two_pass=True)
    rng = np.random.default_rng(seed)
    cap = {"capture_version": np.asarray(CAPTURE_VERSION), "scene": np.asarray(scene_file), "two_mask": two["mask"], "hiz": hizd["data"].numpy()}
    for tag in ("early", "late"):
        cap[f"two_{tag}_visible"] = rng.permutation(two[f"{tag}_visible"])
        tri = two[f"{tag}_indices"].reshape(-1, 3)
        cap[f"two_{tag}_indices"] = tri[rng.permutation(tri.shape[0])].reshape(-1)
    return cap


@pytest.mark.skipif(not CAPTURES, reason="no tests/golden/reference_capture_*.npz yet: needs a Vulkan run of the reference (tools/reference_capture/README.md)")
@pytest.mark.parametrize("path", CAPTURES, ids=[os.path.basename(p) for p in CAPTURES])
def test_checker_against_a_reference_capture(oracle_lib, path):
    cap = dict(np.load(path, allow_pickle=False))
    s, z, hizd, two = _expected(str(cap["scene"]))
    report = compare_capture(cap, s, two, hizd["data"].numpy())
    print(os.path.basename(path), report)


@pytest.mark.gpu
@pytest.mark.skipif(not CAPTURES, reason="no tests/golden/reference_capture_*.npz yet: needs a Vulkan run of the reference (tools/reference_capture/README.md)")
@pytest.mark.parametrize("path", CAPTURES, ids=[os.path.basename(p) for p in CAPTURES])
def test_hip_path_against_a_reference_capture(renderer, oracle_lib, path):
    from oxylus_amd.renderer import ImageAttachment, MainGeometryContext
    from util import gpu_frame

    cap = dict(np.load(path, allow_pickle=False))
    s, z, hizd, _ = _expected(str(cap["scene"]))
    hiz = ImageAttachment.hiz(64, 64, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(torch.from_numpy(z["depth"]).cuda()), hiz))
    got = gpu_frame(renderer, s.to("cuda"), use_hiz=True, hiz=hiz, mask=torch.from_numpy(z["mask_in"]), two_pass=True)
    compare_capture(cap, s, got, hiz.data.cpu().numpy())


def test_comparator_accepts_the_fast_math_envelope_in_any_order(oracle_lib):
    """Self-test: a synthesised capture (fast-math envelope decisions, scrambled lists) passes, and what it reports is what
    tools/unpinned_gap.py measures: differing triangles, all inside the boundary set."""
    cap = _synth_capture(oracle_lib, "pipeline_12x40.npz")
    s, z, hizd, two = _expected("pipeline_12x40.npz")
    report = compare_capture(cap, s, two, hizd["data"].numpy())
    for tag in ("early", "late"):
        assert report[tag]["triangles_differ"] <= report[tag]["boundary_set"]
    assert report["late"]["visible"] + report["early"]["visible"] > 0 and report["hiz_bytes"] > 0


def test_comparator_refuses_a_well_conditioned_difference(oracle_lib):
    """Dropping a triangle OUTSIDE the boundary set, adding a meshlet, flipping a mask bit or a pyramid texel must fail."""
    from capture_compare import boundary_triangle_keys

    s, z, hizd, two = _expected("pipeline_12x40.npz")
    good = _synth_capture(oracle_lib, "pipeline_12x40.npz", seed=2)
    compare_capture(good, s, two, hizd["data"].numpy())
    # (a) a well-conditioned triangle disappears
    tag = "late" if two["late_indices"].size else "early"
    boundary = set(boundary_triangle_keys(s, s.cull_camera(), s.meshlet_instances, np.sort(two[f"{tag}_visible"].view(np.uint32))).tolist())
    tri = good[f"two_{tag}_indices"].reshape(-1, 3)
    keys = tri[:, 0].astype(np.int64) & 0xFFFFFFFF
    victim = next(i for i in range(tri.shape[0]) if int(keys[i]) not in boundary)
    bad = dict(good)
    bad[f"two_{tag}_indices"] = np.delete(tri, victim, axis=0).reshape(-1)
    with pytest.raises(AssertionError, match="NOT in the boundary set"):
        compare_capture(bad, s, two, hizd["data"].numpy())
    # (b) one meshlet more
    bad = dict(good)
    missing = np.setdiff1d(np.arange(s.n_meshlet_instances, dtype=np.uint32), two[f"{tag}_visible"].view(np.uint32))
    bad[f"two_{tag}_visible"] = np.concatenate([good[f"two_{tag}_visible"], missing[:1].view(np.int32)])
    with pytest.raises(AssertionError, match="visible meshlets"):
        compare_capture(bad, s, two, hizd["data"].numpy())
    # (c) a mask bit, (d) a pyramid texel
    bad = dict(good)
    bad["two_mask"] = good["two_mask"].copy()
    bad["two_mask"][0] ^= 1
    with pytest.raises(AssertionError, match="mask"):
        compare_capture(bad, s, two, hizd["data"].numpy())
    bad = dict(good)
    bad["hiz"] = good["hiz"].copy()
    bad["hiz"][5] += 1.0
    with pytest.raises(AssertionError, match="pyramid"):
        compare_capture(bad, s, two, hizd["data"].numpy())
    # (e) a triangle torn apart (corners not adjacent)
    bad = dict(good)
    idx = good[f"two_{tag}_indices"].copy()
    idx[[1, 4]] = idx[[4, 1]]
    bad[f"two_{tag}_indices"] = idx
    with pytest.raises(AssertionError, match="adjacent"):
        compare_capture(bad, s, two, hizd["data"].numpy())
