"""Scene ingestion: the C++ tokenizer / parameter lists / API state machine
against the reference's own parser and api tests (src/core/parser.rs:778-880, src/core/api.rs:979-1045)
and against BASELINE config C0's scene.  CPU only."""
import os

import numpy as np
import pytest

import pbrt_amd
from pbrt_amd import loader
from pbrt_amd._lib import PbrtHipError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C0 = os.path.join(ROOT, "scenes", "c0_check_sphere.pbrt")


def test_tokenizer():  # parser.rs:778-791
    toks, ok = loader.tokenize('Sampler "halton" "integer pixelsamples" 128')
    assert ok and toks == ["Sampler", '"halton"', '"integer pixelsamples"', "128"]
    toks, ok = loader.tokenize('Sampler "128')  # EOF inside a quoted string -> Error::EOF after "Sampler"
    assert not ok and toks == ["Sampler"]
    toks, ok = loader.tokenize('"a\nb"')  # parser.rs:80 UnterminatedString
    assert not ok
    toks, ok = loader.tokenize("Shape [1 2]# trailing comment\n  Next")  # brackets split, comments are tokens
    assert ok and toks == ["Shape", "[", "1", "2", "]", "# trailing comment", "Next"]
    assert loader.tokenize("") == ([], True)


def test_parser_accepts_the_reference_parser_test_input():  # parser.rs:793-800
    ls = loader.load_string('Sampler "halton" "integer pixelsamples" 128')
    assert ls.names["sampler"] == "halton" and ls.spp == (16, 8) and ls.spp[0] * ls.spp[1] == 128


def test_param_lists():  # parser.rs:803-866: the three basic_param_list_entrypoint cases, seen through their effects
    ls = loader.load_string('Camera "perspective" "float fov" 45')
    assert ls.names["camera"] == "perspective" and ls.scene.fov == 45.0
    mesh = ('WorldBegin Shape "trianglemesh" "integer indices" [ 0 1 2 2 3 0 ] "point P" '
            '[-0.5 -0.5 0.5 -0.5 -0.5 -0.5 0.5 -0.5 -0.5 0.5 -0.5 0.5] WorldEnd')
    ls = loader.load_string(mesh)
    assert ls.scene.idx.tolist() == [[0, 1, 2], [2, 3, 0]]
    assert ls.scene.P.tolist() == [[-0.5, -0.5, 0.5], [-0.5, -0.5, -0.5], [0.5, -0.5, -0.5], [0.5, -0.5, 0.5]]
    tex = ('WorldBegin Texture "t" "spectrum" "imagemap"\n "string filename" ["textures/BeoCom.png"]\n'
           ' "float scale" [1.000000]\n "vector v1" [0.500000 0.000000 0.000000]\n WorldEnd')
    ls = loader.load_string(tex)  # parses; image maps are out of scope -> one warning, no error
    assert any("imagemap" in w for w in ls.warnings)


def test_param_errors():  # parser.rs:31-58 Error kinds
    for text, kind in [('Camera "perspective" "float fov" [45 "x"]', "MixedParameters"),
                       ("Camera perspective", "Unquoted"), ("Bogus 1 2 3", "Syntax"), ("LookAt 1 2 x", "Syntax"),
                       ('MakeNamedMedium "a"', "NotImplemented"), ('Include "does/not/exist.pbrt"', "Io"),
                       ('Camera "perspective" "float fov" [45', "Eof"), ("LookAt 1 2 3", "Eof")]:
        with pytest.raises(PbrtHipError) as e:
            loader.load_string(text)
        assert e.value.code == -1 and f" {kind}:" in str(e.value), (text, str(e.value))
    # ... in the reference's own words (the Display strings of parser.rs:31-58)
    for text, words in [("Bogus 1 2 3", "syntax error: 'Bogus'"), ("LookAt 1 2 x", "input not float"), ("Camera perspective", "expected quoted string"),
                        ('MakeNamedMedium "a"', "have not yet implemented 'MakeNamedMedium'"), ("LookAt 1 2 3", "premature EOF"),
                        ('Camera "persp\n', "unterminated string"), ('Camera "perspective" "float fov" [45 "x"]', "mixed string and numeric parameters")]:
        with pytest.raises(PbrtHipError) as e:
            loader.load_string(text)
        assert words in str(e.value), (text, str(e.value))


def test_named_coordinate_systems():  # api.rs:979-1020
    ls = loader.load_string('Identity Scale 2 2 2 CoordinateSystem "two" Identity Scale 3 3 3')
    assert np.array_equal(ls.ctm, np.diag([3, 3, 3, 1]).astype(np.float32))
    ls = loader.load_string('Identity Scale 2 2 2 CoordinateSystem "two" Identity Scale 3 3 3 CoordSysTransform "two"')
    assert np.array_equal(ls.ctm, np.diag([2, 2, 2, 1]).astype(np.float32))
    ls = loader.load_string('Scale 2 2 2 CoordSysTransform "nope"')  # api.rs:727-730: warn, keep the CTM
    assert np.array_equal(ls.ctm, np.diag([2, 2, 2, 1]).astype(np.float32))
    assert "Couldn\u2019t find named coordinate system \"nope\"" in ls.warnings  # api.rs:745, its typographic apostrophe included


def test_attribute_and_transform_stacks():  # api.rs:1022-1045 + :481-522
    ls = loader.load_string("WorldBegin AttributeBegin ActiveTransform StartTime Translate 1 2 3 AttributeEnd Scale 2 2 2")
    assert np.array_equal(ls.ctm, np.diag([2, 2, 2, 1]).astype(np.float32))  # translate popped, active bits restored
    ls = loader.load_string("WorldBegin TransformBegin Translate 1 2 3 TransformEnd Translate 0 0 5")
    assert np.array_equal(ls.ctm[:3, 3], np.array([0, 0, 5], np.float32))
    ls = loader.load_string("WorldBegin AttributeEnd TransformEnd")  # unmatched: logged and ignored (api.rs:497-500)
    assert [w for w in ls.warnings if "Unmatched" in w] == ["Unmatched pbrt.attribute_end() encountered. Ignoring it.",  # api.rs:497
                                                             "Unmatched pbrt.transform_end() encountered. Ignoring it."]  # api.rs:517
    ls = loader.load_string("AttributeBegin")  # verify_world! (api.rs:320-332): outside the world block -> its message, ignored
    assert 'Scene description must be inside world block; "pbrt.attribute_begin" not allowed. Ignoring.' in ls.warnings
    ls = loader.load_string('WorldBegin Camera "perspective"')  # verify_options! (api.rs:304-316)
    assert 'Options cannot be set inside world block; "pbrt.camera" not allowed. Ignoring.' in ls.warnings
    ls = loader.load_string('LightSource "point" WorldBegin PixelFilter "box"')
    assert [w for w in ls.warnings if "Ignoring" in w] == ['Scene description must be inside world block; "pbrt.light_source" not allowed. Ignoring.',
                                                            'Options cannot be set inside world block; "pbrt.pixel_filter" not allowed. Ignoring.']


def test_ctm_ops_match_the_transform_doctests():  # transform.rs:360-443,524-538 through the directives
    ls = loader.load_string("Translate 2 4 6")
    assert np.array_equal(ls.ctm, np.array([[1, 0, 0, 2], [0, 1, 0, 4], [0, 0, 1, 6], [0, 0, 0, 1]], np.float32))
    ls = loader.load_string("Rotate 180 0 0 1")
    c, s = np.float32(np.cos(np.float32(np.pi))), np.float32(np.sin(np.float32(np.pi)))
    assert np.allclose(ls.ctm, np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]), atol=1e-7)
    ls = loader.load_string("Transform [1 0 0 0  0 1 0 0  0 0 1 0  7 8 9 1] ConcatTransform [2 0 0 0 0 2 0 0 0 0 2 0 0 0 0 1]")
    assert np.array_equal(ls.ctm, np.array([[2, 0, 0, 7], [0, 2, 0, 8], [0, 0, 2, 9], [0, 0, 0, 1]], np.float32))


def test_defaults():  # api.rs:231-241 names; pbrt-v3 parameter defaults
    ls = loader.load_string("WorldBegin WorldEnd")
    assert ls.names == {"camera": "perspective", "sampler": "halton", "integrator": "path", "filter": "box",
                        "accelerator": "bvh", "film": "image"}
    assert ls.max_depth == 5 and ls.scene.fov == 90.0 and (ls.scene.xres, ls.scene.yres) == (1280, 720)
    assert ls.scene.idx.shape == (0, 3) and np.array_equal(ls.scene.cam_to_world, np.eye(4, dtype=np.float32))


def _expect_check_sphere(ls, res):
    sd = ls.scene
    assert (sd.xres, sd.yres) == res and sd.fov == 45.0 and ls.spp == (16, 8) and ls.max_depth == 5
    assert ls.filename == "simple.png" and ls.integrator == pbrt_amd.INTEGRATOR_PATH_MIS  # (the file has no Integrator line: "path" as pbrt-v3 means it)
    assert sd.P.tolist() == [[-20, -20, -1], [20, -20, -1], [20, 20, -1], [-20, 20, -1]]  # Translate 0 0 -1 applied
    assert sd.idx.tolist() == [[0, 1, 2], [0, 2, 3]] and sd.mat_id.tolist() == [1, 1]
    assert sd.spheres.tolist() == [[0, 0, 0, 1, 0]]
    assert np.allclose(sd.materials, [[1, .9, .9, .9, 0, 0, 0], [0, .45, .45, .45, 0, 0, 0]])  # mirror; matte mean(tex1, tex2)
    assert sd.lights[0].tolist() == pytest.approx([2, 0, 0, 0, .4, .45, .5])
    d = np.array([-30, 40, 99]) / np.linalg.norm([-30, 40, 99])  # from - to, "point to" defaults to (0, 0, 1) as in pbrt-v3
    assert sd.lights[1, 0] == 1 and np.allclose(sd.lights[1, 1:4], d, atol=1e-6)
    r, g, b = sd.lights[1, 4:7]
    assert 1.0 < r < 1.25 and 0.45 < g < 0.62 and 0.1 < b < 0.25  # 3000 K normalised blackbody x 1.5: warm white
    want_c2w = pbrt_amd.look_at((3, 4, 1.5), (.5, .5, 0), (0, 0, 1))[1]  # camera_to_world = CTM^-1 (api.rs:813-820)
    assert np.array_equal(sd.cam_to_world, want_c2w)
    assert len(ls.warnings) == 1 and "checkerboard" in ls.warnings[0] and "point-sampled" in ls.warnings[0]
    # the ground's Kd is the checkerboard itself: texture 1 on material 1, 8 x 8 checks over the quad's "float st"
    assert sd.mat_tex.tolist() == [0, 1] and sd.textures.shape == (1, 11)
    assert np.allclose(sd.textures[0], [0, .1, .1, .1, .8, .8, .8, 8, 8, 0, 0])
    assert sd.tri_uv.tolist() == [[0, 0, 1, 0, 1, 1], [0, 0, 1, 1, 0, 1]]
    assert ls.sampler == 3  # Sampler "halton": the Halton sampler proper


def test_c0_scene_loads():
    _expect_check_sphere(loader.load_file(C0), (400, 400))


@pytest.mark.skipif(not os.path.exists("/root/reference/scenes/check-sphere.pbrt"), reason="reference tree not mounted")
def test_reference_scene_files_load():
    """The reference's own scene files: its parser stops at line 14 / 20 of this one."""
    _expect_check_sphere(loader.load_file("/root/reference/scenes/check-sphere.pbrt"), (400, 400))
    _expect_check_sphere(loader.load_file("/root/reference/src/core/testdata/scene1.pbrt"), (400, 300))
    ls = loader.load_file("/root/reference/scenes/paramset-lookup.pbrt")
    assert ls.scene.idx.shape[0] == 0


def test_area_light_material_and_orientation():
    text = ('WorldBegin AttributeBegin AreaLightSource "diffuse" "rgb L" [3 2 1] Material "matte" "rgb Kd" [0 0 0] '
            'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0] AttributeEnd '
            'Scale 1 1 -1 Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0] '
            'ReverseOrientation Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0] '
            'LightSource "point" "point from" [1 2 3] "rgb I" [2 2 2] "rgb scale" [.5 .5 .5] WorldEnd')
    sd = loader.load_string(text).scene
    assert sd.materials.tolist() == [[0, 0, 0, 0, 3, 2, 1], [0, .5, .5, .5, 0, 0, 0]]  # area light popped with the attribute
    assert sd.mat_id.tolist() == [0, 1, 1]
    assert sd.idx.tolist() == [[0, 1, 2], [3, 5, 4], [6, 7, 8]]  # mirrored CTM flips winding; ReverseOrientation flips it back
    assert sd.lights.tolist() == [[0, 1, 2, -3, 1, 1, 1]]  # point light through the CTM, I * scale


def test_include(tmp_path):
    (tmp_path / "geo.pbrt").write_text('Shape "sphere" "float radius" 2\n')
    (tmp_path / "main.pbrt").write_text('WorldBegin\nInclude "geo.pbrt"\nWorldEnd\n')
    sd = loader.load_file(tmp_path / "main.pbrt").scene
    assert sd.spheres.tolist() == [[0, 0, 0, 2, 0]]


def test_recursive_include_is_an_error(tmp_path):
    """a file that includes itself must end in a parse error, not in an endless loop."""
    from pbrt_amd import _lib
    f = tmp_path / "self.pbrt"
    f.write_text('Include "self.pbrt"\n')
    with pytest.raises(_lib.PbrtHipError) as e:
        loader.load_file(str(f))
    assert "Include" in str(e.value) and "recursive" in str(e.value)
    a, b = tmp_path / "a.pbrt", tmp_path / "b.pbrt"
    a.write_text('Include "b.pbrt"\n')
    b.write_text('Include "a.pbrt"\n')
    with pytest.raises(_lib.PbrtHipError):
        loader.load_file(str(a))


def test_film_scale_and_filter_radius_leave_the_library():
    """Film "float scale" multiplies every pixel in Film::write_image (film.rs:368-371): the loader exports it; a
    PixelFilter radius reaches the render desc (radii other than 0.5: DESIGN.md 3.11)."""
    ls = loader.load_string('Film "image" "float scale" 2.5 "integer xresolution" 8 "integer yresolution" 8\n'
                            'PixelFilter "box" "float xwidth" 1.5 "float ywidth" 0.5\nWorldBegin\nWorldEnd\n')
    assert ls.film_scale == 2.5
    assert ls.filter_width == (1.5, 0.5) and ls.max_sample_luminance == 0.0
    ls = loader.load_string("WorldBegin\nWorldEnd\n")
    assert ls.film_scale == 1.0 and ls.filter_width == (0.5, 0.5)
    # box.rs:45-55 / api.rs:1058-1064 (test_make_filter): "xwidth" 1 alone -> radius (1, 0.5), inverse radius (1, 2)
    ls = loader.load_string('PixelFilter "box" "float xwidth" 1')
    assert ls.filter_width == (1.0, 0.5) and (1 / ls.filter_width[0], 1 / ls.filter_width[1]) == (1.0, 2.0)
    # Film "float maxsampleluminance" (film.rs:75,279) reaches the render desc too; pbrt-v3's default (infinity) is 0 here
    ls = loader.load_string('Film "image" "float maxsampleluminance" 12.5 "integer xresolution" 8 "integer yresolution" 8')
    assert ls.max_sample_luminance == 12.5 and ls.render_kwargs()["max_sample_luminance"] == 12.5
    film = np.ones((2, 2, 4), np.float32)
    assert np.array_equal(pbrt_amd.film_to_rgb(film, scale=2.5), pbrt_amd.film_to_rgb(film) * np.float32(2.5))


def test_sampler_names():
    """Sampler "halton" (the reference's default name, api.rs:235) and the other low-discrepancy names select the
    (0,2)-sequence sampler, "sobol" the Sobol' sampler with its own dimensions per request;
    "stratified" / "random" / unknown names the stratified one."""
    for name, want in (("halton", 3), ("sobol", 2), ("02sequence", 1), ("lowdiscrepancy", 1), ("stratified", 0), ("random", 0), ("bogus", 0)):
        ls = loader.load_string(f'Sampler "{name}" "integer pixelsamples" 32')
        assert ls.sampler == want, name
        if name != "stratified":
            assert ls.spp[0] * ls.spp[1] == 32
    assert loader.load_string("WorldBegin\nWorldEnd\n").sampler == 3  # no Sampler directive: the default name is "halton" (api.rs:235)
    ls = loader.load_string('Sampler "stratified" "integer xsamples" 3 "integer ysamples" 5')
    assert ls.sampler == 0 and ls.spp == (3, 5)


def test_integrator_mis_switch():
    """`Integrator "path"` in a scene file means pbrt-v3's path integrator -- the reference's default name (api.rs:239), whose direct-light
    estimate is multiple-importance-sampled: integrator 2 (the parser's default since round 6, also when the file has no
    Integrator line).  "bool mis" "false" selects SURVEY A8's estimator (integrator 0, what BASELINE's synthetic configs name through the C
    ABI).  The variants combine freely with a wide box filter, the Halton sampler and textures (render_kernel_x): nothing is dropped,
    nothing warned about."""
    assert loader.load_string('Integrator "path"').integrator == 2 and loader.load_string("WorldBegin WorldEnd").integrator == 2
    ls = loader.load_string('Integrator "path" "bool mis" "false"')
    assert ls.integrator == 0 and not ls.warnings
    ls = loader.load_string('Integrator "path" "bool mis" "true" "integer maxdepth" 7')
    assert ls.integrator == 2 and ls.max_depth == 7 and not ls.warnings
    assert loader.load_string('Integrator "directlighting" "bool mis" "true"').integrator == 1
    ls = loader.load_string('PixelFilter "box" "float xwidth" 2 "float ywidth" 2\nSampler "halton"\nIntegrator "path" "bool mis" "true"\nWorldBegin\nWorldEnd\n')
    assert ls.integrator == 2 and ls.sampler == 3 and ls.filter_width == (2.0, 2.0) and not ls.warnings


# ---- data formats on the input side of the path that the reference stops short of (parser.rs:283-300: NotImplemented) ----

def _ply_bytes(fmt, verts, faces, uv=None, with_normals=False, extra_element=False, index_type="int", count_type="uchar",
               coord_type="float", uv_names=("u", "v"), face_first=False):
    """A PLY file as pbrt-v3 scenes hold them: fmt in ascii / binary_little_endian / binary_big_endian."""
    import struct
    e = "<" if fmt != "binary_big_endian" else ">"
    code = {"float": "f", "double": "d", "int": "i", "uint": "I", "uchar": "B", "short": "h", "ushort": "H", "char": "b"}
    head = ["ply", f"format {fmt} 1.0", "comment made by the test"]
    vprops = [(coord_type, n) for n in "xyz"]
    if with_normals:
        vprops += [("float", n) for n in ("nx", "ny", "nz")]
    if uv is not None:
        vprops += [("float", uv_names[0]), ("float", uv_names[1])]
    vhead = [f"element vertex {len(verts)}"] + [f"property {t} {n}" for t, n in vprops]
    fhead = [f"element face {len(faces)}", f"property list {count_type} {index_type} vertex_indices", "property uchar flags"]
    xhead = ["element edge 2", "property int a", "property list uchar short tags"] if extra_element else []
    order = [fhead, xhead, vhead] if face_first else [vhead, xhead, fhead]
    head += [l for part in order for l in part] + ["end_header"]
    rows = {"v": [], "f": [], "x": []}
    for i, p in enumerate(verts):
        r = [(coord_type, c) for c in p]
        if with_normals:
            r += [("float", 0.0), ("float", 0.0), ("float", 1.0)]
        if uv is not None:
            r += [("float", uv[i][0]), ("float", uv[i][1])]
        rows["v"].append(r)
    for f in faces:
        rows["f"].append([(count_type, len(f))] + [(index_type, k) for k in f] + [("uchar", 7)])
    if extra_element:
        rows["x"] = [[("int", 5), ("uchar", 2), ("short", -1), ("short", 9)], [("int", 6), ("uchar", 0)]]
    body_rows = [rows[k] for k in (("f", "x", "v") if face_first else ("v", "x", "f"))]
    if fmt == "ascii":
        text = "\n".join(head) + "\n"
        for part in body_rows:
            for r in part:
                text += " ".join(repr(float(v)) if t in ("float", "double") else str(v) for t, v in r) + "\n"
        return text.encode()
    out = ("\n".join(head) + "\n").encode()
    for part in body_rows:
        for r in part:
            for t, v in r:
                out += struct.pack(e + code[t], v)
    return out


_PLY_VERTS = [(-1.0, -1.0, 0.0), (1.0, -1.0, 0.0), (1.0, 1.0, 0.5), (-1.0, 1.0, 0.25), (0.0, 2.0, 1.0)]
_PLY_UV = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0), (0.5, 2.0)]
_PLY_FACES = [(0, 1, 2, 3), (3, 2, 4), (0, 1, 2, 3, 4)]  # a quad, a triangle, a pentagon (ignored, as pbrt-v3 does)


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
@pytest.mark.parametrize("variant", ["plain", "uv+normals", "st+extra", "double+uint", "face_first"])
def test_plymesh_loads_like_the_same_trianglemesh(tmp_path, fmt, variant):
    """Shape "plymesh" (pbrt-v3 plymesh.cpp): vertices, optional (u, v) under any of its four spellings, triangles and quads -- a quad
    becomes (0 1 2) (3 0 2) --, other properties and elements skipped; the arrays are those of the equivalent "trianglemesh", through the
    CTM, in all three PLY encodings."""
    kw = {"plain": {}, "uv+normals": dict(uv=_PLY_UV, with_normals=True), "st+extra": dict(uv=_PLY_UV, uv_names=("s", "t"), extra_element=True),
          "double+uint": dict(coord_type="double", index_type="uint", count_type="int", uv=_PLY_UV, uv_names=("texture_u", "texture_v")),
          "face_first": dict(face_first=True, extra_element=True)}[variant]
    (tmp_path / "geometry").mkdir()
    (tmp_path / "geometry" / "m.ply").write_bytes(_ply_bytes(fmt, _PLY_VERTS, _PLY_FACES, **kw))
    tex = 'Texture "c" "spectrum" "checkerboard" "float uscale" 4 "float vscale" 4 Material "matte" "texture Kd" "c"'
    pre = f'WorldBegin {tex} Translate 1 2 3 Scale 2 2 2 '
    (tmp_path / "s.pbrt").write_text(pre + 'Shape "plymesh" "string filename" "geometry/m.ply" WorldEnd')
    flat_p = " ".join(str(c) for v in _PLY_VERTS for c in v)
    flat_uv = " ".join(str(c) for v in _PLY_UV for c in v)
    uvp = f'"float uv" [{flat_uv}]' if "uv" in kw else ""
    ref = loader.load_string(pre + f'Shape "trianglemesh" "integer indices" [0 1 2 3 0 2 3 2 4] "point P" [{flat_p}] {uvp} WorldEnd')
    ls = loader.load_file(str(tmp_path / "s.pbrt"))
    assert ls.scene.idx.tolist() == [[0, 1, 2], [3, 0, 2], [3, 2, 4]]
    assert np.array_equal(ls.scene.P, ref.scene.P) and np.array_equal(ls.scene.idx, ref.scene.idx)
    assert np.array_equal(ls.scene.tri_uv, ref.scene.tri_uv) and np.array_equal(ls.scene.mat_id, ref.scene.mat_id)
    assert ls.scene.P[2].tolist() == [3.0, 4.0, 4.0]  # (1, 1, 0.5) * 2 + (1, 2, 3)
    assert any("1 faces with other than 3 or 4 vertices" in w for w in ls.warnings)
    assert not any("not used" in w for w in ls.warnings)


def test_plymesh_bad_files_are_skipped_with_a_warning(tmp_path):
    """pbrt-v3 logs a PLY it cannot read and goes on without the shape; so does this parser -- whatever the file holds."""
    good = _ply_bytes("binary_little_endian", _PLY_VERTS, _PLY_FACES)
    cases = {
        "missing.ply": None,
        "magic.ply": b"plx" + good[3:],
        "truncated.ply": good[:-9],
        "huge.ply": good.replace(b"element vertex 5", b"element vertex 4000000000"),
        "range.ply": _ply_bytes("ascii", _PLY_VERTS, [(0, 1, 7)]),
        "negative.ply": _ply_bytes("ascii", _PLY_VERTS, [(0, 1, -2)]),
        "noxyz.ply": good.replace(b"property float z", b"property float w"),
        "format.ply": good.replace(b"binary_little_endian", b"binary_middle_endian"),
        "nohdr.ply": good[:good.index(b"end_header")],
        "empty.ply": b"",
    }
    for name, data in cases.items():
        if data is not None:
            (tmp_path / name).write_bytes(data)
        ls = loader.load_string(f'WorldBegin Shape "plymesh" "string filename" "{name}" WorldEnd', base_dir=str(tmp_path))
        assert ls.scene.idx.shape[0] == 0 and any("plymesh" in w and "skipped" in w for w in ls.warnings), (name, ls.warnings)
    ls = loader.load_string('WorldBegin Shape "plymesh" WorldEnd')
    assert any("filename" in w for w in ls.warnings)


def test_object_instancing_flattens_into_the_scene():
    """ObjectBegin / ObjectEnd / ObjectInstance (pbrt-v3 api.cpp): an object's shapes carry their own CTM, an instance adds the CTM of its
    ObjectInstance on top; ObjectBegin / ObjectEnd push and pop the graphics state; area lights inside an object do not emit; a mirroring
    instance keeps the geometric normal on the transformed side."""
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]'
    text = ('WorldBegin Material "mirror" ObjectBegin "o" Material "matte" "rgb Kd" [.1 .2 .3] Translate 0 0 1 ' + tri +
            ' Shape "sphere" "float radius" 0.5 ObjectEnd ' + tri +  # after ObjectEnd: the mirror and the identity CTM are back
            ' AttributeBegin Translate 10 0 0 Scale 2 2 2 ObjectInstance "o" AttributeEnd'
            ' AttributeBegin Scale -1 1 1 ObjectInstance "o" AttributeEnd ObjectInstance "nope" WorldEnd')
    ls = loader.load_string(text)
    sd = ls.scene
    assert sd.idx.shape[0] == 3 and sd.spheres.shape[0] == 2
    assert sd.P[sd.idx[0]].tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0]]                 # the scene's own triangle, not moved
    assert sd.P[sd.idx[1]].tolist() == [[10, 0, 2], [12, 0, 2], [10, 2, 2]]             # ((p + (0,0,1)) * 2) + (10,0,0)
    assert sd.P[sd.idx[2]].tolist() == [[0, 0, 1], [0, 1, 1], [-1, 0, 1]]               # mirrored in x: corners 1 and 2 change places
    n = np.cross(sd.P[sd.idx[2]][1] - sd.P[sd.idx[2]][0], sd.P[sd.idx[2]][2] - sd.P[sd.idx[2]][0])
    assert n.tolist() == [0, 0, 1]                                                          # the normal (0,0,1) maps to (0,0,1) under Scale -1 1 1
    assert [int(sd.materials[m, 0]) for m in sd.mat_id.tolist()] == [1, 0, 0]  # mirror outside, the object's matte inside
    assert np.allclose(sd.materials[sd.mat_id[1], 1:4], [.1, .2, .3])
    assert sd.spheres[:, :4].tolist() == [[10, 0, 2, 1.0], [0, 0, 1, 0.5]]     # centre (0,0,1) r 0.5: scaled by 2 and moved; mirrored
    assert [int(sd.materials[int(m), 0]) for m in sd.spheres[:, 4]] == [0, 0]
    assert any('Unable to find instance named "nope"' in w for w in ls.warnings)
    # states pbrt-v3 reports and survives
    for text, msg in [('WorldBegin ObjectBegin "a" ObjectBegin "b" ObjectEnd WorldEnd', "inside of instance definition"),
                      ('WorldBegin ObjectEnd WorldEnd', "outside of instance definition"),
                      ('WorldBegin ObjectBegin "a" ObjectInstance "a" ObjectEnd WorldEnd', "can't be called inside instance definition"),
                      ('WorldBegin ObjectBegin "a" ' + tri + ' WorldEnd', "Missing end to ObjectBegin"),
                      ('WorldBegin ObjectBegin "a" AreaLightSource "diffuse" ' + tri + ' ObjectEnd ObjectInstance "a" WorldEnd', "Area lights not supported with object instancing"),
                      ('ObjectBegin "a"', "world block")]:
        ls = loader.load_string(text)
        assert any(msg in w for w in ls.warnings), (text, ls.warnings)
    ls = loader.load_string('WorldBegin ObjectBegin "a" AreaLightSource "diffuse" "rgb L" [5 5 5] ' + tri + ' ObjectEnd ObjectInstance "a" WorldEnd')
    assert ls.scene.idx.shape[0] == 1 and not ls.scene.materials[:, 4:7].any()
    ls = loader.load_string('WorldBegin ObjectBegin "a" ' + tri + ' WorldEnd')  # an unfinished object is not part of the scene
    assert ls.scene.idx.shape[0] == 0


def test_object_instances_cannot_ask_for_unbounded_memory(monkeypatch):
    """A file of n bytes can ask for ~n^2 triangles (instances x the object's triangles): beyond 2^24 (what pbrt_hip_scene_create takes) the file is refused with an error
    instead of an allocation.  (The bound is lowered here: reaching the real one takes most of a gigabyte of host arrays.)"""
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]'
    text = 'WorldBegin ObjectBegin "o" ' + " ".join([tri] * 10) + " ObjectEnd " + 'ObjectInstance "o" ' * 11 + "WorldEnd"
    assert loader.load_string(text).scene.idx.shape[0] == 110
    monkeypatch.setenv("PBRT_HIP_MAX_SCENE_TRIANGLES", "100")
    with pytest.raises(PbrtHipError) as e:
        loader.load_string(text)
    assert "2^24" in str(e.value)


def test_textured_sphere_keeps_its_texture():
    """A checkerboard named as a sphere's matte Kd stays a checkerboard (the sphere's own (u, v)); a rotated or mirrored
    sphere is rendered with (u, v) about the world's axes and says so."""
    tex = 'Texture "c" "spectrum" "checkerboard" "float uscale" 8 "float vscale" 4 Material "matte" "texture Kd" "c" '
    ls = loader.load_string("WorldBegin " + tex + 'Translate 1 2 3 Scale 2 2 2 Shape "sphere" "float radius" 0.5 WorldEnd')
    assert ls.scene.mat_tex[int(ls.scene.spheres[0, 4])] == 1 and ls.scene.textures.shape[0] == 1
    assert ls.scene.spheres[0, :4].tolist() == [1, 2, 3, 1.0]
    assert not [w for w in ls.warnings if "sphere" in w]
    for xf in ("Rotate 30 1 0 0", "Scale -1 1 1"):
        ls = loader.load_string("WorldBegin " + tex + xf + ' Shape "sphere" WorldEnd')
        assert any("world's axes" in w for w in ls.warnings), (xf, ls.warnings)


def test_camera_parameters_the_pinhole_ignores_are_reported():
    ls = loader.load_string('Camera "perspective" "float fov" 30 "float lensradius" 0.1 "float focaldistance" 4 "float screenwindow" [-1 1 -1 1]')
    assert any("lensradius" in w for w in ls.warnings) and any("screenwindow" in w for w in ls.warnings)
    ls = loader.load_string('Camera "perspective" "float fov" 30 "float lensradius" 0 "float shutteropen" 0')
    assert not ls.warnings


def test_integrator_parameters_without_effect_are_reported():
    ls = loader.load_string('Integrator "path" "float rrthreshold" 0.5 "string lightsamplestrategy" "spatial"')
    assert any("rrthreshold" in w for w in ls.warnings) and any("lightsamplestrategy" in w for w in ls.warnings)
    ls = loader.load_string('Integrator "directlighting" "string strategy" "all"')
    assert any('"strategy" "all"' in w for w in ls.warnings)
    ls = loader.load_string('Integrator "path" "float rrthreshold" 1 "string lightsamplestrategy" "uniform" "integer maxdepth" 7')
    assert not ls.warnings and ls.max_depth == 7


def test_log_messages_are_the_references_where_it_has_them():
    """api.rs:692 (unknown light), :939 (unknown spectrum texture), :897-901 (animated transforms: the start transform is used)."""
    ls = loader.load_string('WorldBegin LightSource "laser" Material "matte" "texture Kd" "nothing" '
                            'ActiveTransform EndTime Translate 1 0 0 ActiveTransform All Shape "sphere" LightSource "point" WorldEnd')
    assert "light_source: light type 'laser' unknown." in ls.warnings
    assert "Spectrum texture 'nothing' is unknown" in ls.warnings
    assert 'Animated transformations set; ignoring for "pbrt.shape" and using the start transform only' in ls.warnings
    assert 'Animated transformations set; ignoring for "pbrt.light_source" and using the start transform only' in ls.warnings
    assert ls.scene.spheres[0, :3].tolist() == [0, 0, 0] and ls.scene.lights[0, 1:4].tolist() == [0, 0, 0]  # the start transform


def test_find_one_returns_the_first_value():
    """paramset.rs:237-513: find_one_float / _int / _bool / _string return the FIRST value of a parameter (`pl.0.first()`), the default
    when the parameter is absent or empty -- the reference's rule, not pbrt-v3's "exactly one value"."""
    ls = loader.load_string('Camera "perspective" "float fov" [45 50]\nFilm "image" "integer xresolution" [320 1] "integer yresolution" [] "string filename" ["a.png" "b.png"]')
    assert ls.scene.fov == 45.0 and ls.scene.xres == 320 and ls.scene.yres == 720 and ls.filename == "a.png"
    ls = loader.load_string('Integrator "path" "bool mis" ["true" "false"] "integer maxdepth" [3 9]')
    assert ls.integrator == 2 and ls.max_depth == 3


def test_mutated_scene_files_load_or_are_refused_through_the_abi():
    """4 000 mutations of two scene files (tools/parser_fuzz.py: tokens inserted, stretches deleted / duplicated, truncation, numbers turned into
    0 / -1 / 2^32 / 2^24 + 1 / 1e30): each either loads or comes back as PbrtHipError -- an error code and a message through the C ABI; no
    exception of another kind, no crash (1.2 M files: profiles/r06u_parser_fuzz.txt).  A loaded file's arrays are consistent with each other; VALUES are
    pbrt_hip_scene_create's to refuse (a resolution of 0; `LookAt 1e30 0 5 ...` loads with a camera matrix of NaNs as it would in the reference: "scene_create: camera
    matrix is not finite")."""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parser_fuzz
    from pbrt_amd._lib import PbrtHipError
    ok, err = parser_fuzz.run(1, 4000)
    assert ok > 400 and err > 400 and ok + err == 4000
    rnd = random.Random(2)
    for _ in range(1500):
        try:
            sd = loader.load_string(parser_fuzz.mutate(rnd.choice(parser_fuzz.texts), rnd)).scene
        except PbrtHipError:
            continue
        n = sd.idx.shape[0]
        assert sd.mat_id.shape[0] == n and (n == 0 or int(sd.idx.max()) < sd.P.shape[0]) and sd.tri_uv.shape[0] in (0, n)
        assert (n == 0 or int(sd.mat_id.max()) < sd.materials.shape[0]) and (sd.spheres.shape[0] == 0 or int(sd.spheres[:, 4].max()) < sd.materials.shape[0])
