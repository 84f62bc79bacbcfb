"""CPU: librt_core.so loads and exports every symbol include/*.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    from raytracing_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared("rt_core.h") + _declared("rt_host.h")
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_lists_match_headers():
    from raytracing_amd import _lib
    from raytracing_amd.scene import HOST_EXPORTS
    assert sorted(_lib.EXPORTS) == _declared("rt_core.h")
    assert sorted(HOST_EXPORTS) == _declared("rt_host.h")


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device the core refuses to start (skipped where a GPU exists)."""
    import pytest
    from raytracing_amd import ComputeShader
    from raytracing_amd.core import RtError
    try:
        cs = ComputeShader()
    except RtError as e:
        assert e.status == 1 and "no HIP device" in str(e)
    else:
        cs.Delete()
        pytest.skip("a GPU is present")


def test_struct_layouts_are_std430():
    from oracle import gltf_flatten as G
    assert (G.VERTEX_DT.itemsize, G.MATERIAL_DT.itemsize, G.TRIANGLE_DT.itemsize, G.NODE_DT.itemsize, G.LIGHT_DT.itemsize) == (80, 80, 128, 48, 48)
    assert G.TRIANGLE_DT.fields["meshIndex"][1] == 108 and G.TRIANGLE_DT.fields["materialIndex"][1] == 112
    assert G.NODE_DT.fields["isLeaf"][1] == 28 and G.NODE_DT.fields["children"][1] == 32
    assert G.MATERIAL_DT.fields["ior"][1] == 64 and G.LIGHT_DT.fields["lightType"][1] == 28
