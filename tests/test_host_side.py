"""Host-side pieces of the product that need no GPU: the C-ABI library loads and exports every
symbol include/tiny_renderer.h declares, the loaders agree with independent decoders, the
prepares agree with the oracle, and the synthetic scene / instancing helpers are well formed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import helpers as H


def test_library_exports_every_declared_symbol(built):
    import tiny_renderer_amd as T
    from tiny_renderer_amd import _lib
    hdr = open(os.path.join(H.REPO, "include", "tiny_renderer.h")).read()
    declared = set(re.findall(r"\b(tr_[a-z0-9_]+)\s*\(", hdr))
    lib = C.CDLL(T.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = T.load_library()
    assert L.tr_abi_version() == 3
    names = [L.tr_pipeline_name(i).decode() for i in range(L.tr_pipeline_count())]
    assert tuple(names) == T.PIPELINES  # shader.rs:100-109


def test_scene_creation_fails_loudly_without_a_gpu(built, small_synthetic):
    import torch
    import tiny_renderer_amd as T
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    mesh, texs = small_synthetic
    with pytest.raises(T.TinyRendererError) as e:
        T.Scene(64, 64, mesh, texs, "phong")
    assert e.value.code == -6  # TR_E_HIP: no CPU fallback exists


def test_obj_loader_matches_python_reader(built):
    import tiny_renderer_amd as T
    for name, counts in (("diablo", (2519, 3263, 2519, 5022)), ("african_head", (1258, 1339, 1258, 2492))):
        d = H.asset_dir(name)
        if d is None:
            pytest.skip("reference assets not available")
        got = T.load_obj(os.path.join(d, "model.obj"))
        ref = H.load_obj_py(os.path.join(d, "model.obj"))
        assert (got["pos"].shape[0], got["tex"].shape[0], got["nrm"].shape[0], got["idx"].shape[0]) == counts
        for k in ("pos", "tex", "nrm", "idx"):
            assert np.array_equal(got[k], ref[k]), k


def test_tga_loader_matches_pil(built):
    import hashlib
    import tiny_renderer_amd as T
    # sha256 of the decoded RGB bytes (PIL, this container)
    prints = {("diablo", "texture.tga"): "90fe242b874ec5fe", ("diablo", "normal_map.tga"): "b7a91e2d456b12fb",
              ("diablo", "normal_map_tangent.tga"): "1ef1cb262f58bb9b", ("diablo", "specular_map.tga"): "67f36a6fbfa240de",
              ("african_head", "texture.tga"): "883953448607a493", ("african_head", "normal_map.tga"): "0d99ece19adc680e",
              ("african_head", "normal_map_tangent.tga"): "08fd0d67cca56823",
              ("african_head", "specular_map.tga"): "abcd8711b75c6ed9"}
    for (name, f), digest in prints.items():
        d = H.asset_dir(name)
        if d is None:
            pytest.skip("reference assets not available")
        got = T.load_tga(os.path.join(d, f))
        assert got.shape == (1024, 1024, 3)
        assert hashlib.sha256(got.tobytes()).hexdigest()[:16] == digest, (name, f)
        assert np.array_equal(got, H.load_tga_pil(os.path.join(d, f)))


def test_tga_variants_roundtrip(built, tmp_path):
    """Types 1 / 2 / 3 / 9 / 10 / 11, 8 / 24 / 32 bpp, both origins: written by hand, decoded by the
    loader; and the TGA frame writer against the loader."""
    import tiny_renderer_amd as T
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    img[2, 1:5] = img[2, 1]  # a run for the RLE packets

    def header(typ, bpp, desc):
        return bytes([0, 0, typ, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7, 0, 5, 0, bpp, desc])

    def rle(rows, bpp):
        out = bytearray()
        px = [bytes(p) for row in rows for p in row]
        i = 0
        while i < len(px):
            j = i
            while j + 1 < len(px) and px[j + 1] == px[i] and j - i < 127:
                j += 1
            if j > i:
                out += bytes([0x80 | (j - i)]) + px[i]
                i = j + 1
            else:
                out += bytes([0]) + px[i]
                i += 1
        return bytes(out)

    bgr = img[..., ::-1]
    bgra = np.concatenate([bgr, np.full((5, 7, 1), 200, np.uint8)], -1)
    grey = img[..., :1]
    cases = [(2, 24, 0x00, bgr[::-1].tobytes(), img), (2, 24, 0x20, bgr.tobytes(), img),
             (2, 32, 0x28, bgra.tobytes(), img), (10, 24, 0x00, rle(bgr[::-1], 3), img),
             (10, 32, 0x08, rle(bgra[::-1], 4), img), (3, 8, 0x00, grey[::-1].tobytes(), grey.repeat(3, -1)),
             (11, 8, 0x20, rle(grey, 1), grey.repeat(3, -1))]
    for k, (typ, bpp, desc, body, want) in enumerate(cases):
        p = tmp_path / ("t%d.tga" % k)
        p.write_bytes(header(typ, bpp, desc) + body)
        assert np.array_equal(T.load_tga(str(p)), want), (typ, bpp, desc)
    # colour-mapped (types 1 / 9): 8-bit indices into a 24- or 32-bit palette with a first-entry offset
    pal = rng.integers(0, 256, (11, 3), dtype=np.uint8)            # rgb
    ind = rng.integers(0, 11, (5, 7, 1), dtype=np.uint8)
    ind[3, 2:6] = ind[3, 2]
    for k, (typ, bits, first, desc) in enumerate([(1, 24, 0, 0x20), (1, 32, 3, 0x00), (9, 24, 5, 0x20)]):
        entries = pal[:, ::-1] if bits == 24 else np.concatenate([pal[:, ::-1], np.full((11, 1), 9, np.uint8)], 1)
        hd = bytes([0, 1, typ, first & 0xFF, first >> 8, 11, 0, bits, 0, 0, 0, 0, 7, 0, 5, 0, 8, desc])
        rows = (ind + first).astype(np.uint8)
        rows = rows if desc & 0x20 else rows[::-1]
        body = rle(rows, 1) if typ == 9 else rows.tobytes()
        p = tmp_path / ("m%d.tga" % k)
        p.write_bytes(hd + entries.tobytes() + body)
        assert np.array_equal(T.load_tga(str(p)), pal[ind[..., 0]]), (typ, bits, first)
    # the frame writer: what it writes, the loader reads back
    out = tmp_path / "frame.tga"
    T._lib.check(T.load_library().tr_save_tga_rgb8(str(out).encode(), img.ctypes.data, 7, 5))
    assert np.array_equal(T.load_tga(str(out)), img)
    bad = tmp_path / "bad.tga"
    bad.write_bytes(header(2, 16, 0))
    with pytest.raises(T.TinyRendererError) as e:
        T.load_tga(str(bad))
    assert e.value.code == -8
    bad.write_bytes(bytes([0, 1, 1, 0, 0, 2, 0, 24, 0, 0, 0, 0, 1, 0, 1, 0, 8, 0]) + bytes(6) + bytes([7]))  # index past the palette
    with pytest.raises(T.TinyRendererError) as e:
        T.load_tga(str(bad))
    assert e.value.code == -8
    with pytest.raises(T.TinyRendererError) as e:
        T.load_tga(str(tmp_path / "missing.tga"))
    assert e.value.code == -7


def test_obj_edge_cases(built, tmp_path):
    import tiny_renderer_amd as T
    p = tmp_path / "m.obj"
    p.write_text("# comment\nv 0 0 0\nv 1 0 0 1.0\nv 0 1 0\nv 1 1 0\nvt 0.5 0.25\nvt 0 1 0\nvn 0 0 1\n"
                 "g grp\nusemtl x\nf 1/1/1 2/2/1 3/1/1 4/2/1\nf -1/-1/-1 -2/-2/-1 -3/-1/-1\n")
    m = T.load_obj(str(p))
    assert m["pos"].shape == (4, 3) and m["tex"].shape == (2, 3) and m["idx"].shape == (2, 9)
    assert list(m["idx"][0]) == [0, 0, 0, 1, 1, 0, 2, 0, 0]      # first three of the quad (scene.rs:224-226)
    assert list(m["idx"][1]) == [3, 1, 0, 2, 0, 0, 1, 1, 0]      # negative = relative indices
    assert m["tex"][0, 2] == 0.0
    q = tmp_path / "pn.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    with pytest.raises(T.TinyRendererError) as e:
        T.load_obj(str(q))
    assert e.value.code == -3   # not a PTN polygon: the reference panics at scene.rs:218


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_prepare_matches_oracle(built, kind):
    """shader.rs:183-279: the product's host prepares against the oracle's, bit for bit."""
    import tiny_renderer_amd as T
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for trial in range(40):
        ca, la = rng.uniform(-3, 3, 2)
        cam, lt = H.camera(ca), H.light(la)
        if trial % 4 == 3:   # general positions, not just the app's orbit
            cam = (list(rng.uniform(-2, 2, 3)), list(rng.uniform(-0.2, 0.2, 3)), [0.1, 1.0, -0.2])
            lt = list(rng.uniform(-1, 1, 3))
        W, Hh = int(rng.integers(16, 5000)), int(rng.integers(16, 5000))
        uo, up = O.Uniforms(), T._lib.Uniforms()
        if kind == 2:   # pass 2 inherits shadow_matrix from pass 1
            O.prepare(1, W, Hh, lt, *cam, uniforms=uo)
            T.prepare_uniforms(1, W, Hh, lt, *cam, uniforms=up)
        eo, uo = O.prepare(kind, W, Hh, lt, *cam, uniforms=uo)
        ep, up = T.prepare_uniforms(kind, W, Hh, lt, *cam, uniforms=up)
        assert (eo == 0) == (ep == 0)
        for name, _ in O.Uniforms._fields_:
            a = np.array(getattr(uo, name), np.float32).view(np.uint32)
            b = np.array(getattr(up, name), np.float32).view(np.uint32)
            assert np.array_equal(a, b), (kind, trial, name)


def test_synthetic_scene_and_instancing(built):
    import tiny_renderer_amd as T
    mesh, texs = T.synthetic_scene()
    assert mesh["idx"].shape == (5022, 9) and all(t.shape == (1024, 1024, 3) for t in texs)
    assert mesh["tex"][:, :2].min() >= 0.001 and mesh["tex"][:, :2].max() <= 0.999
    m2, t2 = T.synthetic_scene()
    assert np.array_equal(m2["pos"], mesh["pos"]) and np.array_equal(t2[0], texs[0])  # deterministic
    g = T.instanced_grid(mesh, 8)
    assert g["idx"].shape == (5022 * 64, 9) and g["pos"].shape[0] == 64 * mesh["pos"].shape[0]
    assert g["idx"][:, 0::3].max() < g["pos"].shape[0] and np.abs(g["pos"][:, :2]).max() <= 1.0


def test_band_rows_partition(built):
    """tr_band_rows: disjoint, ordered, covering; equal bands when the rank count divides the height."""
    import tiny_renderer_amd as T
    for height, n in ((4096, 8), (4096, 1), (800, 3), (7, 7), (1080, 8), (8192, 4)):
        rows = [T.band_rows(height, n, r) for r in range(n)]
        assert rows[0][0] == 0 and rows[-1][1] == height
        for a, b in zip(rows, rows[1:]):
            assert a[1] == b[0]
        assert all(r1 > r0 for r0, r1 in rows)
        if height % n == 0:
            assert len({r1 - r0 for r0, r1 in rows}) == 1
    with pytest.raises(T.TinyRendererError):
        T.band_rows(4, 8, 0)
    with pytest.raises(T.TinyRendererError):
        T.band_rows(64, 4, 4)


def test_specular_exact_is_checked_on_this_host(built):
    """tr_specular_exact() compares the shipped powf with the running C library's at first use."""
    import tiny_renderer_amd as T
    assert T.load_library().tr_specular_exact() == 1  # this image and the GPU box: glibc 2.35, same tables


def test_png_writer_roundtrip(built, tmp_path):
    """tr_save_png_rgb8 (stored deflate blocks, own CRC-32 / Adler-32) read back by an independent decoder."""
    from PIL import Image
    import tiny_renderer_amd as T
    rng = np.random.default_rng(4)
    for shape in ((1, 1), (37, 53), (300, 21846)):   # the last: rows longer than one 65535-byte stored block
        img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
        path = str(tmp_path / "t.png")
        T.save_png(path, img)
        assert np.array_equal(np.array(Image.open(path).convert("RGB")), img)
    with pytest.raises(T.TinyRendererError):
        T.save_png(str(tmp_path / "no" / "dir.png"), img)


def test_cli_ranks_are_started_as_a_module(tmp_path):
    """`--gpus N` starts its ranks with torch.distributed.run as `-m tiny_renderer_amd.cli` (a plain script path has
    no parent package: its relative imports fail in every rank) with the package's parent on PYTHONPATH, from any
    working directory.  Here one rank is started the way launch_ranks builds the command, elsewhere, and must get
    past its imports (no GPU needed: the hook returns before anything touches one)."""
    import os
    import subprocess
    import sys
    from tiny_renderer_amd.sharded import rank_command, rank_environment
    cmd = rank_command(2, "tiny_renderer_amd.cli", ["--synthetic", "--gpus", "2"], 29517)
    assert "-m" in cmd and cmd[cmd.index("-m", 3) + 1] == "tiny_renderer_amd.cli"
    env = rank_environment()
    env.update(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", TR_CLI_IMPORT_CHECK="1")
    module_part = cmd[cmd.index("-m", 3):]          # what torch.distributed.run executes per rank
    r = subprocess.run([sys.executable] + module_part, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "rank 1 of 2: imports ok" in r.stdout
