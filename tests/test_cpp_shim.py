"""The C++ surface an engine would include -- ox::amd::RendererInstance, oxylus_amd/host/RendererInstance.hpp -- driven with real
buffers by a compiled C++ program (tests/cpp/shim_frame.cpp) in the order of RendererInstance::render's 3D pass
(Oxylus/src/Render/RendererInstance.cpp:793-884); every output is byte-compared with the committed fixture (plain pipeline) and
with the checker run live (two-pass occlusion in the reference's order: early cull -> new pyramid -> late cull)."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

import oracle
from oxylus_amd import lib as L
from oxylus_amd.synth import hiz_layout, make_depth

from util import oracle_hiz, scene_from_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_container(path, items):
    n = len(items)
    off = 8 + n * 40
    head = [b"OXCF", struct.pack("<I", n)]
    for name, data in items:
        head.append(struct.pack("<24sQQ", name.encode(), off, len(data)))
        off += len(data)
    with open(path, "wb") as f:
        f.write(b"".join(head))
        for _, data in items:
            f.write(data)


def _read_container(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"OXCF"
    (n,) = struct.unpack_from("<I", raw, 4)
    out = {}
    for i in range(n):
        name, off, size = struct.unpack_from("<24sQQ", raw, 8 + i * 40)
        out[name.rstrip(b"\0").decode()] = raw[off:off + size]
    return out


def _image_desc(w, h, levels, offs, total):
    return struct.pack("<IIII13QQ", w, h, levels, 0, *(list(offs) + [0] * (13 - len(offs))), total)


def build_shim_frame(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    L.build()
    exe = str(tmp_path / "shim_frame")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "oxylus_amd", "host"), os.path.join(ROOT, "tests", "cpp", "shim_frame.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "oxylus_amd"), "-loxcull", "-Wl,-rpath," + os.path.join(ROOT, "oxylus_amd"), "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_frame_program_builds(tmp_path):
    """(no GPU) the C++ driver compiles against the shim header + C ABI with plain g++ and links to liboxcull.so."""
    exe = build_shim_frame(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stdout


@pytest.mark.gpu
def test_cpp_shim_runs_the_reference_sequence_with_data(tmp_path, oracle_lib):
    exe = build_shim_frame(tmp_path)
    s, z = scene_from_golden(os.path.join(ROOT, "tests", "golden", "pipeline_12x40.npz"), "cpu")
    t = s._lod_tables
    Lc = s.spec.lod_count
    raw = lambda x: x.contiguous().numpy().tobytes()  # noqa: E731
    lods, meshes = s.lods.clone(), s.meshes.clone()
    lods[:, 0:5] = 0  # the program fills the pointers in for ITS device allocations
    meshes[:, 0] = 0
    meshes[:, 4] = 0
    relocs = []

    def reloc(field_section, field_offset, target, addend):
        relocs.append(struct.pack("<24s24sQQ", field_section.encode(), target.encode(), int(field_offset), int(addend)))

    for i in range(lods.shape[0]):  # GPU::MeshLOD (SceneGPU.hpp:125-139): meshlets, meshlet_bounds, local_triangle_indices, indirect_vertex_indices
        reloc("lods", i * 64 + 8, "meshlets", int(t["meshlet_start"][i]) * 16)
        reloc("lods", i * 64 + 16, "bounds", int(t["meshlet_start"][i]) * 16)
        reloc("lods", i * 64 + 24, "micro", int(t["micro_start"][i]))
        reloc("lods", i * 64 + 32, "vidx", int(t["vidx_start"][i]) * 4)
    for m in range(meshes.shape[0]):  # GPU::Mesh (SceneGPU.hpp:141-152): vertex_positions, lods
        reloc("meshes", m * 64 + 0, "positions", int(t["mesh_vertex_start"][m]) * 8)
        reloc("meshes", m * 64 + 32, "lods", m * 64 * Lc)
    depth0 = torch.from_numpy(z["depth"]).contiguous()
    depth1 = make_depth(depth0.shape[1], depth0.shape[0], 20, seed=71)
    depth1 = (depth1 + torch.rand(depth1.shape, generator=torch.Generator().manual_seed(71)) * 1e-4).contiguous()
    hw = 64
    levels, offs, total = hiz_layout(hw, hw)
    cam = s.cull_camera()
    N = s.n_meshlet_instances
    items = [(name, raw(getattr(s, name))) for name in ("bounds", "meshlets", "micro", "vidx", "positions", "transforms", "mesh_instances")]
    items += [("lods", raw(lods)), ("meshes", raw(meshes)), ("depth0", raw(depth0)), ("depth1", raw(depth1)), ("mask_in", z["mask_in"].tobytes()),
              ("reloc", b"".join(relocs)), ("camera", bytes(cam)), ("hizdesc", _image_desc(hw, hw, levels, offs, total)),
              ("depth0desc", _image_desc(depth0.shape[1], depth0.shape[0], 1, [0], depth0.numel() * 4)),
              ("depth1desc", _image_desc(depth1.shape[1], depth1.shape[0], 1, [0], depth1.numel() * 4)), ("max_meshlets", struct.pack("<I", N))]
    fin, fout = str(tmp_path / "in.oxcf"), str(tmp_path / "out.oxcf")
    _write_container(fin, items)
    p = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr + p.stdout
    got = _read_container(fout)
    u32 = lambda b: np.frombuffer(b, dtype=np.uint32)  # noqa: E731

    # ---- sequence A against the committed fixture (plain pipeline with cull_meshes)
    ca = u32(got["A_counters"])
    assert ca[0] == int(z["plain_total"]) and ca[3] == int(z["plain_cull_meshlets_cmd_x"])
    assert np.array_equal(u32(got["A_meshlet_instances"]).reshape(-1, 2), z["plain_meshlet_instances"].view(np.uint32))
    assert np.array_equal(u32(got["A_visible"]), z["plain_visible"].view(np.uint32))
    assert np.array_equal(u32(got["A_indices"]), z["plain_indices"].view(np.uint32))
    assert np.array_equal(u32(got["A_mesh_instances"]).reshape(-1, 5)[:, 1], z["plain_lod_index"].view(np.uint32))
    # ... and the draw that consumes them (draw_for_visbuffer) against the committed raster fixture + its two resolves
    vd = np.load(os.path.join(ROOT, "tests", "golden", "raster_512x384.npz"))["visdepth"]
    assert np.array_equal(np.asarray([cam.projection_view[i] for i in range(16)], dtype=np.float32), np.asarray(z["camera_pv"], dtype=np.float32).reshape(-1))
    assert np.array_equal(np.frombuffer(got["A_visdepth"], dtype=np.int64).reshape(384, 512), vd)
    assert np.array_equal(u32(got["A_draw_depth"]).reshape(384, 512), (vd >> 32).astype(np.uint32))
    assert np.array_equal(u32(got["A_draw_vis"]).reshape(384, 512), (vd & 0xFFFFFFFF).astype(np.uint32))

    # ---- sequence B against the checker, run here in the same order
    sc = s.clone()
    mli, _ = oracle.cull_meshes(sc, cam, L.CULL_TEST_ALL)
    hz0, lv, of = oracle_hiz(depth0, hw, hw)
    assert got["B_hiz0"] == hz0.numpy().tobytes()
    v = oracle.Visibility(mli.shape[0], 0, 0)
    out = torch.zeros(max(mli.shape[0], 1), dtype=torch.int32)
    mask = torch.from_numpy(z["mask_in"].copy())
    n_e = oracle.cull_meshlets_hiz(sc, cam, mli, L.CULL_TEST_ALL, oracle.make_hiz(hz0, hw, hw, lv, of), v, mask, out)
    assert np.array_equal(u32(got["B_early_visible"]), out[:n_e].numpy().view(np.uint32))
    assert np.array_equal(u32(got["B_early_indices"]), oracle.cull_triangles(sc, cam, mli, out, 0, n_e).numpy().view(np.uint32))
    hz1, _, _ = oracle_hiz(depth1, hw, hw)
    assert got["B_hiz1"] == hz1.numpy().tobytes()
    n_l = oracle.cull_meshlets_hiz(sc, cam, mli, L.CULL_TEST_ALL | L.CULL_LATE_PASS, oracle.make_hiz(hz1, hw, hw, lv, of), v, mask, out)
    assert np.array_equal(u32(got["B_late_visible"]), out[v.early:v.early + n_l].numpy().view(np.uint32))
    assert np.array_equal(u32(got["B_late_indices"]), oracle.cull_triangles(sc, cam, mli, out, v.early, n_l).numpy().view(np.uint32))
    assert got["B_mask"] == mask.numpy().tobytes()
    cb = u32(got["B_late_counters"])
    assert (cb[0], cb[1], cb[2]) == (mli.shape[0], v.early, v.late) and n_e > 0 and n_l > 0
