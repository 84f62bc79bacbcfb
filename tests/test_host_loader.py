"""CPU: the product's C++ GLTF loader + flatten + light marshalling (rth_load_scene, dry run without a GPU)
against the independent numpy restatement in oracle/gltf_flatten.py, byte for byte (pads masked, Q17)."""
import numpy as np
import pytest

from oracle import gltf_flatten as G

SLOTS = {0: ("vertices", G.VERTEX_DT), 2: ("materials", G.MATERIAL_DT), 4: ("triangles", G.TRIANGLE_DT), 5: ("lights", G.LIGHT_DT)}


def _same(a, b):
    for f in b.dtype.names:
        x, y = a[f], b[f]
        if x.dtype == np.float32:
            x, y = np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32)
        if not np.array_equal(x, y):
            return False
    return True


@pytest.mark.parametrize("scene", [2, 3, 4, 7])
def test_scene_buffers_match_oracle(oracle_lib, models_root, scene):
    from raytracing_amd import scene as S
    S.Scene.LoadScene(None, scene, models_root)
    flat = G.load_scene(models_root, scene)
    for slot, (key, dt) in SLOTS.items():
        got = np.frombuffer(S.Scene.buffer(slot), dt)
        assert len(got) == len(flat[key]), key
        assert _same(got, flat[key]), key
    m = np.frombuffer(S.Scene.buffer(1), np.float32).reshape(-1, 4, 4)
    assert np.array_equal(m, flat["matrices"])
    nodes = np.frombuffer(S.Scene.buffer(3), G.NODE_DT)
    assert _same(nodes, oracle_lib.build_bvh(flat["triangles"]))


def test_load_models_explicit_list(models_root):
    from raytracing_amd import scene as S
    S.Scene.LoadModels(None, [models_root + "/box/box.gltf"], [((0, 6.75254, 0), (0, 0, 0), 0, 10.0)])
    flat = G.flatten([models_root + "/box/box.gltf"])
    got = np.frombuffer(S.Scene.buffer(4), G.TRIANGLE_DT)
    assert _same(got, flat["triangles"])
    lights = np.frombuffer(S.Scene.buffer(5), G.LIGHT_DT)
    assert len(lights) == 1 and lights["lightType"][0] == 0 and np.isnan(lights["dir"][0]).all()   # normalize(0) = NaN


# Scene 1 of the reference's table (Scene.cpp:28-31) without its first model, whose base colour is a progressive JPEG
# (no decoder here): five models, two point lights; the cube is normal-mapped, the glass sphere transmissive.
SCENE1_MODELS = ["cube/cube.gltf", "sphere1/sphere.gltf", "glassSphere/sphere.gltf", "walls/walls.gltf", "roughSphere/roughSphere.gltf"]
SCENE1_LIGHTS = [((2, 5, 0), (0, 0, 0), 0, 50.0), ((-2, 5, 0), (0, 0, 0), 0, 50.0)]


def test_five_model_scene_buffers_match_oracle(oracle_lib, models_root):
    """Several models in one scene: vertex / mesh / material offsets of Model::BVH (model.cpp:562-582), texture handles in
    load order, normal-map material (cube) next to untextured glass."""
    from raytracing_amd import scene as S
    paths = [models_root + "/" + m for m in SCENE1_MODELS]
    S.Scene.LoadModels(None, paths, SCENE1_LIGHTS)
    flat = G.flatten(paths)
    flat["lights"], flat["hasSun"] = G.lights_to_gpu(SCENE1_LIGHTS)
    for slot, (key, dt) in SLOTS.items():
        got = np.frombuffer(S.Scene.buffer(slot), dt)
        assert len(got) == len(flat[key]), key
        assert _same(got, flat[key]), key
    assert np.array_equal(np.frombuffer(S.Scene.buffer(1), np.float32).reshape(-1, 4, 4), flat["matrices"])
    assert _same(np.frombuffer(S.Scene.buffer(3), G.NODE_DT), oracle_lib.build_bvh(flat["triangles"]))
    m = flat["materials"]
    assert m["hasNormalTexture"][0] == 1 and m["hasBaseTexture"][0] == 1 and m["transmissionFactor"][2] == 1


def test_material_quirks(models_root):
    """ior defaults to -1 without KHR_materials_ior, metallicFactor to 1 when absent (model.cpp:266-276)."""
    from raytracing_amd import scene as S
    S.Scene.LoadScene(None, 4, models_root)
    m = np.frombuffer(S.Scene.buffer(2), G.MATERIAL_DT)
    assert len(m) == 5
    assert (m["ior"] == -1).all() and (m["transmissionFactor"] == 0).all()
    assert m["metallicFactor"][0] == 0 and (m["metallicFactor"][1:] == 1).all()
    S.Scene.LoadScene(None, 7, models_root)
    m = np.frombuffer(S.Scene.buffer(2), G.MATERIAL_DT)
    assert m["transmissionFactor"][3] == 1 and abs(m["ior"][3] - 1.45) < 1e-6


def test_png_decoder_matches_pil(oracle_lib, models_root):
    from raytracing_amd import scene as S
    for rel in ("box/Untitled.png", "walls/Untitled.png", "cube/normal_mapping_normal_map.png", "sphere1/baseColor.png"):
        assert np.array_equal(S.decode_png(models_root + "/" + rel), oracle_lib.decode_texture(models_root + "/" + rel))


def test_missing_file_is_an_error(models_root):
    from raytracing_amd import scene as S
    from raytracing_amd.core import RtError
    with pytest.raises(RtError):
        S.Scene.LoadScene(None, 6, models_root)   # Dragon.bin is absent from the reference checkout


def test_camera_helpers_match_oracle():
    from raytracing_amd import scene as S
    for f in (0, 1, 59, 119):
        pos, ori = S.orbit_camera(f)
        wp, wo = G.orbit_camera(f)
        assert np.allclose(pos, wp, rtol=0, atol=2e-6) and np.allclose(ori, wo, rtol=0, atol=2e-7)
