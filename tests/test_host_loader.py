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
    # the failed load leaves nothing behind: the next scene loads (a Model that threw in its constructor used to stay
    # registered and was freed twice here)
    S.Scene.LoadScene(None, 4, models_root)
    assert len(np.frombuffer(S.Scene.buffer(4), G.TRIANGLE_DT)) == 4802


def test_camera_helpers_match_oracle():
    from raytracing_amd import scene as S
    for f in (0, 1, 59, 119):
        pos, ori = S.orbit_camera(f)
        wp, wo = G.orbit_camera(f)
        assert np.allclose(pos, wp, rtol=0, atol=2e-6) and np.allclose(ori, wo, rtol=0, atol=2e-7)


def test_exported_frame_uses_the_camera_of_the_previous_update():
    """Renderer::NextFrame encodes frame f, calls UpdateCamera and only then increments currentFrame (Renderer.cpp:151,164),
    and main.cpp dispatches before NextFrame (main.cpp:138,198): exported frame 0 shows the user's camera, frame 1
    UpdateCamera with currentFrame = 0 (x = -PI/6), frame f >= 1 the orbit step f - 1; orbit step 119 is never rendered."""
    import math
    from raytracing_amd import render_path
    from raytracing_amd import scene as S
    assert render_path.frame_camera(0) == S.DEFAULT_CAMERA
    pos, ori = render_path.frame_camera(1)
    x = np.float32(0.0) / np.float32(120.0) - np.float32(3.14159265359) / np.float32(6.0)
    assert np.allclose(pos, (math.sin(x) * 15.0, 3.5, math.cos(x) * 15.0), rtol=0, atol=2e-6)
    assert render_path.frame_camera(1) == S.orbit_camera(0)
    for f in (2, 60, 119):
        assert render_path.frame_camera(f) == S.orbit_camera(f - 1)
        wp, wo = G.orbit_camera(f - 1)
        assert np.allclose(render_path.frame_camera(f)[0], wp, rtol=0, atol=2e-6)


def test_node_transforms_children_and_the_q12_walk(oracle_lib, models_root, tmp_path):
    """A GLTF with what none of the bundled files has: TRS and `matrix` node transforms, `children`, several nodes.
    Model::traverseNode (model.cpp:74-147) composes matrix * node.matrix * T * R * S down the hierarchy and, after a
    node's children, also walks node + 1 from the identity (Q12) — so meshes load more than once.  The C++ loader and
    the numpy restatement must agree byte for byte on all six buffers, matrices included (glm's quaternion -> matrix)."""
    import json
    import shutil
    from raytracing_amd import scene as S
    src = models_root + "/scene4"                       # 16 distinct meshes, 5 materials
    dst = tmp_path / "nodes"
    shutil.copytree(src, dst)
    j = json.load(open(dst / "scene4.gltf"))
    c, s = 0.9914449, 0.1305262            # 15 degrees about x as a unit quaternion (x, y, z, w)
    # three nodes: the walk loads meshes 0, 1 (child of 0), 2 (Q12 from 1), then 1 and 2 again (Q12 from 0).  (Two copies
    # of a mesh fit one leaf; five identical triangles would send the reference's SAH split into its unbounded recursion.)
    j["nodes"] = [
        {"mesh": 0, "translation": [1.5, 0.25, -2.0], "rotation": [s, 0.0, 0.0, c], "scale": [1.25, 0.5, 2.0], "children": [1]},
        {"mesh": 1, "matrix": [0.0, 0.0, -1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, -3.0, 0.5, 1.0, 1.0], "translation": [0.0, 1.0, 0.0]},
        {"mesh": 2, "scale": [0.5, 0.5, 0.5], "rotation": [0.0, 0.3826834, 0.0, 0.9238795]},
    ]
    # Q16: the material comes from JSON["meshes"][position in the walk], not from the mesh the node names
    for k, m in enumerate(j["meshes"]):
        m["primitives"][0]["material"] = (k * 2) % 5
    json.dump(j, open(dst / "scene4.gltf", "w"))
    paths = [str(dst / "scene4.gltf")]
    lights = [((0, 6.75254, 0), (0, 0, 0), 0, 10.0)]
    S.Scene.LoadModels(None, paths, lights)
    flat = G.flatten(paths)
    assert sorted(set(flat["triangles"]["materialIndex"].tolist())) == [0, 1, 2, 3, 4]       # positions 0..4 -> materials 0, 2, 4, 1, 3
    assert len(flat["matrices"]) == 5                       # nodes 0, 1 (child of 0), 2 (Q12 from 1), 1 and 2 again (Q12 from 0)
    got_m = np.frombuffer(S.Scene.buffer(1), np.float32).reshape(-1, 4, 4)
    assert got_m.shape == flat["matrices"].shape
    assert np.array_equal(got_m.view(np.uint32), np.ascontiguousarray(flat["matrices"], np.float32).view(np.uint32))
    assert not np.array_equal(got_m[0], np.eye(4, dtype=np.float32))
    for slot, (key, dt) in SLOTS.items():
        if key == "lights":
            continue
        got = np.frombuffer(S.Scene.buffer(slot), dt)
        assert len(got) == len(flat[key]), key
        assert _same(got, flat[key]), key
    assert _same(np.frombuffer(S.Scene.buffer(3), G.NODE_DT), oracle_lib.build_bvh(flat["triangles"]))
