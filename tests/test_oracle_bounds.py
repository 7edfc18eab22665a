"""CPU known-answer tests of the bounds-producer oracle.  The reference has no tests for
AssetManager_GLTF.cpp:573-578,683-744 and meshoptimizer v1.2 is not vendored, so the answers below are derived by
hand from the published definitions (meshopt_quantizeHalf / meshopt_quantizeSnorm / meshopt_computeClusterBounds)."""
import numpy as np
import torch

import oracle
from oxylus_amd.synth import build_meshlets_simple, make_mesh


def test_quantize_half_known_answers():
    q = oracle.quantize_half
    assert q(0.0) == 0x0000 and q(-0.0) == 0x8000
    assert q(1.0) == 0x3C00 and q(-2.0) == 0xC000 and q(0.5) == 0x3800
    assert q(65504.0) == 0x7BFF          # largest finite half
    assert q(65519.9) == 0x7BFF          # just below the rounding boundary of the largest finite half
    assert q(65520.0) == 0x7C00          # rounds up into infinity
    assert q(1.0e9) == 0x7C00 and q(-1.0e9) == 0xFC00
    assert q(float("inf")) == 0x7C00 and q(float("-inf")) == 0xFC00
    assert q(float("nan")) & 0x7FFF == 0x7E00
    assert q(2.0 ** -14) == 0x0400       # smallest normal half
    assert q(2.0 ** -14 * 0.999) == 0    # would be a denormal half: flushed
    assert q(-6.0e-5) == 0x8000
    assert q(1.0 + 2.0 ** -11) == 0x3C01  # exact tie: away from zero (IEEE round-to-even gives 0x3C00)
    assert q(1.0 + 3 * 2.0 ** -11) == 0x3C02  # tie, both rules agree
    assert q(0.1) == 0x2E66


def test_quantize_half_matches_ieee_off_the_special_cases():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(20000) * 10.0, rng.uniform(-60000, 60000, 5000), rng.uniform(-1e-3, 1e-3, 5000)]).astype(np.float32)
    want = x.astype(np.float16).view(np.uint16)
    got = np.array([oracle.quantize_half(float(v)) for v in x], dtype=np.uint16)
    f = np.abs(x.astype(np.float64))
    # where the rules differ: exact ties (13 low mantissa bits == 0x1000) and magnitudes below 2^-14
    tie = (x.view(np.uint32) & 0x1FFF) == 0x1000
    ok = (f >= 2.0 ** -14) & ~tie
    assert np.array_equal(got[ok], want[ok])
    assert np.all(got[f < 2.0 ** -14] & 0x7FFF == 0)


def test_quantize_half_round_trip_of_every_half():
    from oracle import lib
    for h in range(0, 0x7C01):  # every non-negative finite half and +inf
        f = lib().orc_dequantize_half(h)
        want = h if h >= 0x0400 or h == 0 else 0  # denormal halfs decode to 0
        assert oracle.quantize_half(f) == want, hex(h)


def test_quantize_snorm_known_answers():
    q = oracle.quantize_snorm
    assert [q(v, 8) for v in (0.0, 1.0, -1.0, 2.0, -3.0)] == [0, 127, -127, 127, -127]
    assert q(0.5, 8) == 64 and q(-0.5, 8) == -64          # 63.5 + 0.5 -> 64 (truncation after the signed half)
    assert q(0.0039, 8) == 0 and q(0.00394, 8) == 1       # 0.4953 / 0.50038 before the +0.5
    assert q(1.0, 10) == 511 and q(-1.0, 10) == -511


def _bounds_of(positions, tris):
    positions = torch.tensor(positions, dtype=torch.float32)
    meshlets, vidx, micro = build_meshlets_simple(torch.tensor(tris, dtype=torch.int64))
    b, mesh6, qpos = oracle.build_meshlet_bounds(positions, meshlets, vidx, micro)
    return b.numpy().view(np.uint16), mesh6.numpy(), qpos.numpy().view(np.uint16)


def _s8(word, hi):
    v = (int(word) >> (8 if hi else 0)) & 0xFF
    return v - 256 if v > 127 else v


def test_single_triangle_cone_and_aabb():
    # one CCW triangle in the z = 2 plane: normal (0,0,1), mindp = 1, cutoff = sqrt(1 - 1) = 0,
    # axis_s8 = (0,0,127), quantisation error 0, cutoff_s8 = int(127 * 0 + 1) = 1
    b, mesh6, qpos = _bounds_of([[0, 0, 2], [4, 0, 2], [0, 2, 2]], [[0, 1, 2]])
    h = oracle.quantize_half
    assert list(b[0][:3]) == [h(2.0), h(1.0), h(2.0)]          # center = (max + min) / 2
    assert list(b[0][4:7]) == [h(4.0), h(2.0), h(0.0)]          # extent = max - min
    assert (_s8(b[0][3], False), _s8(b[0][3], True), _s8(b[0][7], False), _s8(b[0][7], True)) == (0, 0, 127, 1)
    assert np.array_equal(mesh6, np.array([2, 1, 2, 4, 2, 0], dtype=np.float32))
    assert list(qpos[1]) == [h(4.0), h(0.0), h(2.0), 0]


def test_opposite_normals_give_a_degenerate_cone():
    # two triangles facing +z and -z: mindp = -1 <= 0.1 -> cutoff_s8 = 127 and the axis stays zero
    b, _, _ = _bounds_of([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1]], [[0, 1, 2], [3, 5, 4]])
    assert (_s8(b[0][3], False), _s8(b[0][3], True), _s8(b[0][7], False), _s8(b[0][7], True)) == (0, 0, 0, 127)


def test_only_degenerate_triangles_leave_cone_data_zero_but_keep_the_aabb():
    b, mesh6, _ = _bounds_of([[1, 1, 1], [3, 1, 1], [5, 1, 1]], [[0, 1, 1], [0, 1, 2]])  # repeated corner, collinear
    assert (_s8(b[0][3], False), _s8(b[0][3], True), _s8(b[0][7], False), _s8(b[0][7], True)) == (0, 0, 0, 0)
    h = oracle.quantize_half
    assert list(b[0][:3]) == [h(3.0), h(1.0), h(1.0)] and list(b[0][4:7]) == [h(4.0), h(0.0), h(0.0)]
    assert np.array_equal(mesh6, np.array([3, 1, 1, 4, 0, 0], dtype=np.float32))


def test_two_normals_axis_is_their_bisector():
    # normals (0,0,1) and (1,0,0): the two points are the sphere's seed pair, centre = midpoint,
    # axis = (1,0,1)/sqrt(2), mindp = cos 45 deg, cutoff = sin 45 deg; s8 axis = 90 (0.70711 * 127 + 0.5 = 90.3)
    b, _, _ = _bounds_of([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], [[0, 1, 2], [0, 2, 3]])
    ax = (_s8(b[0][3], False), _s8(b[0][3], True), _s8(b[0][7], False))
    assert ax == (90, 0, 90)
    e = abs(90 / 127 - 2 ** -0.5) * 2
    assert _s8(b[0][7], True) == int(127 * (2 ** -0.5 + e) + 1)


def test_cone_is_conservative_on_real_meshes():
    for kind in ("sphere", "terrain"):
        pos, tris = make_mesh(kind, n=20, seed=5)
        meshlets, vidx, micro = build_meshlets_simple(tris)
        b, _, _ = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
        bw = b.numpy().view(np.uint16)
        P = pos.numpy().astype(np.float64)
        narrow = 0
        for m, (vo, to, vc, tc) in enumerate(meshlets.tolist()):
            cut = _s8(bw[m][7], True) / 127.0
            if cut >= 1.0:
                continue
            narrow += 1
            axis = np.array([_s8(bw[m][3], False), _s8(bw[m][3], True), _s8(bw[m][7], False)]) / 127.0
            for t in range(tc):
                i = [int(vidx[vo + int(micro[to + 3 * t + k])]) for k in range(3)]
                n = np.cross(P[i[1]] - P[i[0]], P[i[2]] - P[i[0]])
                if np.linalg.norm(n) == 0:
                    continue
                n /= np.linalg.norm(n)
                # every normal is within the cone: angle(n, axis) <= asin(cutoff) (+ quantisation slack already inside cutoff)
                assert np.dot(n, axis) >= np.sqrt(max(0.0, 1.0 - cut * cut)) * np.linalg.norm(axis) - 2e-2
        assert narrow > 0
