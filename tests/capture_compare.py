"""Comparator between a REFERENCE CAPTURE -- buffers dumped from a Vulkan run of oxylusengine/Oxylus on a scene exported by
tools/reference_capture/export_scene.py -- and the checker (oracle/) or the HIP path.  Test infrastructure: imports oracle/.

What can be compared how:
  * the reference allocates output slots with atomics: every list is compared as a SORTED SET, never in order;
  * the pyramid is mins and point samples of the depth image (hiz.slang): byte for byte;
  * meshlet decisions (frustum / cone / occlusion / LOD) do not move under the rewrites a fast-math shader compiler may apply
    (0-1 of 1 M in every scene measured): the visible sets must be EQUAL, and so must the mask;
  * triangle decisions sit on a cancelling determinant: the two triangle sets may differ, but only inside the checker's
    conditioning-aware boundary set (orc_triangle_boundary_flags: |det - 1e-4| and clip.z within 4 * 2^-24 * sum |products|).
A capture that passes pins the oracle to the Vulkan path on that scene; until one exists the oracle says "parity unpinned"."""
from __future__ import annotations

import numpy as np
import torch

import oracle

CAPTURE_VERSION = 1
# keys of a capture .npz (u32 lists as int32 / uint32, any order inside a list)
REQUIRED = ("capture_version", "scene", "two_early_visible", "two_late_visible", "two_early_indices", "two_late_indices", "two_mask")


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a)).view(np.uint32).reshape(-1) if np.asarray(a).dtype.itemsize == 4 else np.asarray(a, dtype=np.uint32).reshape(-1)


def _triangles(indices, corner_bits: int) -> np.ndarray:
    """Packed index list -> sorted unique triangle keys (meshlet instance << bits) | 3t; checks that corners come in threes."""
    u = _u32(indices)
    assert u.size % 3 == 0, "index count is not a multiple of 3"
    t = u.reshape(-1, 3)
    assert np.all(t[:, 1] == t[:, 0] + 1) and np.all(t[:, 2] == t[:, 0] + 2), "a triangle's three packed indices are not adjacent (visbuffer.slang:13-14)"
    keys = np.sort(t[:, 0])
    assert np.all(np.diff(keys) > 0), "a triangle was emitted twice"
    return keys


def boundary_triangle_keys(scene, cam, meshlet_instances, visible_sorted: np.ndarray, corner_bits: int = 8) -> np.ndarray:
    """Keys of the triangles of the visible meshlets whose decision a legal re-association / fusion may flip."""
    vis = torch.from_numpy(np.ascontiguousarray(visible_sorted.astype(np.int32)))
    flags = oracle.triangle_boundary_flags(scene, cam, meshlet_instances, vis, 0, vis.numel()).numpy()
    s, t = np.nonzero(flags)
    return np.sort((visible_sorted.astype(np.uint64)[s] << corner_bits | (3 * t).astype(np.uint64)).astype(np.uint32))


def compare_pass(tag: str, scene, cam, meshlet_instances, got_visible, got_indices, want_visible, want_indices, corner_bits: int = 8) -> dict:
    """One cull_geometry call of the capture (`got`) against the checker / HIP path (`want`).  Raises AssertionError with what differs."""
    gv, wv = np.sort(_u32(got_visible)), np.sort(_u32(want_visible))
    assert gv.size == wv.size, f"{tag}: {gv.size} visible meshlets in the capture, {wv.size} expected"
    assert np.array_equal(gv, wv), f"{tag}: visible meshlet sets differ in {int(np.setxor1d(gv, wv).size)} ids (a meshlet decision moved: not a rounding matter)"
    gt, wt = _triangles(got_indices, corner_bits), _triangles(want_indices, corner_bits)
    diff = np.setxor1d(gt, wt)
    boundary = boundary_triangle_keys(scene, cam, meshlet_instances, wv, corner_bits)
    outside = np.setdiff1d(diff, boundary)
    assert outside.size == 0, (f"{tag}: {outside.size} of the {diff.size} differing triangles are NOT in the boundary set (first: instance {int(outside[0]) >> corner_bits}, "
                               f"triangle {(int(outside[0]) & ((1 << corner_bits) - 1)) // 3}): a well-conditioned decision differs")
    return {"visible": int(gv.size), "triangles_capture": int(gt.size), "triangles_expected": int(wt.size), "triangles_differ": int(diff.size),
            "boundary_set": int(boundary.size)}


def compare_capture(capture: dict, scene, expected: dict, hiz_expected: np.ndarray = None) -> dict:
    """capture: the arrays of a reference_capture_*.npz; expected: oracle_frame / gpu_frame result of the two-pass HiZ frame on the same
    scene, mask and pyramid (keys early_visible, late_visible, early_indices, late_indices, mask)."""
    for k in REQUIRED:
        assert k in capture, f"capture lacks '{k}' (tools/reference_capture/README.md)"
    assert int(capture["capture_version"]) == CAPTURE_VERSION
    cam = scene.cull_camera()
    report = {}
    for tag in ("early", "late"):
        report[tag] = compare_pass(tag, scene, cam, scene.meshlet_instances, capture[f"two_{tag}_visible"], capture[f"two_{tag}_indices"],
                                   expected[f"{tag}_visible"], expected[f"{tag}_indices"])
    assert np.array_equal(_u32(capture["two_mask"]), _u32(expected["mask"])), "visibility mask after the late pass differs"
    if "hiz" in capture and hiz_expected is not None:
        assert np.array_equal(_u32(capture["hiz"]), _u32(hiz_expected)), "HiZ pyramid differs (mins and point samples only: must be byte-identical)"
        report["hiz_bytes"] = int(_u32(capture["hiz"]).size * 4)
    return report
