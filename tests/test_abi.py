"""The C-ABI library loads and exports every symbol include/oxcull.h (the drop-in boundary) and include/oxcull_debug.h (test / harness /
measurement hooks, not part of the boundary) declare (no GPU needed)."""
import ctypes
import os
import re

from oxylus_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(headers=("oxcull.h", "oxcull_debug.h")):
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(oxc_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(L.EXPORTS)


def test_the_boundary_header_carries_no_debug_hooks():
    """include/oxcull.h is what an engine binds: the oxc_debug_* / profiling / probe entry points live in oxcull_debug.h, and nothing the
    reference-named C++ surface (oxylus_amd/host/RendererInstance.hpp) or the integration notes bind comes from there."""
    boundary = _declared_symbols(("oxcull.h",))
    debug = _declared_symbols(("oxcull_debug.h",))
    assert not [n for n in boundary if n.startswith("oxc_debug_") or n.startswith("oxc_profile_") or n == "oxc_stream_read_probe"]
    assert debug and all(n.startswith("oxc_debug_") or n.startswith("oxc_profile_") or n == "oxc_stream_read_probe" for n in debug)
    shim = open(os.path.join(ROOT, "oxylus_amd", "host", "RendererInstance.hpp")).read()
    assert "oxcull_debug.h" not in shim and not [n for n in debug if n in shim]


def test_library_exports_every_declared_symbol():
    L.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"liboxcull.so does not export {name}"
    assert lib.oxc_abi_version() == L.ABI_VERSION == 5


def test_struct_sizes_match_reference_layouts():
    # GPU::CullCamera is a 96-byte push constant (SceneGPU.hpp:222-229)
    assert ctypes.sizeof(L.CullCamera) == 96
    assert ctypes.sizeof(L.Buffer) == 16
    assert ctypes.sizeof(L.Image) == 24 + 13 * 8
    assert ctypes.sizeof(L.Counters) == 24
    assert ctypes.sizeof(L.KernelTimes) == 16 * 8 + 16 * 4 + 8


def test_context_struct_matches_the_header(tmp_path):
    """sizeof / offsetof of the ctypes mirror == what a C compiler makes of include/oxcull.h."""
    import subprocess

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "oxcull_debug.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(oxc_cull_geometry_context), '
                   'offsetof(oxc_cull_geometry_context, small_triangle_cull), offsetof(oxc_cull_geometry_context, visibility_buffer), sizeof(oxc_kernel_times)); return 0; }\n')
    exe = str(tmp_path / "sz")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    size, off_small, off_vis, kt = [int(x) for x in subprocess.check_output([exe]).split()]
    assert size == ctypes.sizeof(L.CullGeometryContext)
    assert off_small == L.CullGeometryContext.small_triangle_cull.offset
    assert off_vis == L.CullGeometryContext.visibility_buffer.offset
    assert kt == ctypes.sizeof(L.KernelTimes)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "oxylus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "oxcull_oracle" not in src, f"{f} references the oracle"


def test_cpp_shim_compiles_and_links(tmp_path):
    """oxylus_amd/host/RendererInstance.hpp (the reference-named C++ surface) builds against the
    C ABI with plain g++ and links to liboxcull.so."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        import pytest

        pytest.skip("no g++")
    L.build()
    exe = str(tmp_path / "shim_check")
    subprocess.check_call(["g++", "-std=c++20", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oxylus_amd", "host"),
                           os.path.join(ROOT, "oxylus_amd", "host", "shim_compile_check.cpp"), "-o", exe, "-L" + os.path.join(ROOT, "oxylus_amd"),
                           "-loxcull", "-Wl,-rpath," + os.path.join(ROOT, "oxylus_amd"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe]).decode()
    assert "shim links" in out


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/oxcull.h must compile as strict C99 (what a cgo / JNI / ctypes-gen binding would parse)."""
    import subprocess

    src = tmp_path / "c99.c"
    src.write_text('#include "oxcull.h"\n#include "oxcull_debug.h"\nint main(void) { oxc_cull_geometry_context c; oxc_mesh_blob_layout l; oxc_kernel_times k; (void)c; (void)l; (void)k; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), "-fsyntax-only", str(src)])
