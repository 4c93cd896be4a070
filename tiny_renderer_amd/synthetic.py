"""Procedural stand-in scene for boxes without the reference's assets,
and the 8x8 instancing rule of BASELINE.json's configs[4]."""
import numpy as np


def _splitmix64(seed, n):
    x = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _noise_image(seed, size, smooth=8):
    """Deterministic low-pass noise in [0,1), float32 [size,size,3]."""
    with np.errstate(over="ignore"):
        r = _splitmix64(seed, size * size * 3)
    a = ((r >> np.uint64(40)).astype(np.float64) / float(1 << 24)).reshape(size, size, 3)
    for axis in (0, 1):
        acc = np.zeros_like(a)
        for k in range(smooth):
            acc += np.roll(a, k, axis=axis)
        a = acc / smooth
    return a.astype(np.float32)


def synthetic_scene(n_lat=31, n_lon=81, tex_size=1024, radius=0.8):
    """UV sphere with 2*n_lat*n_lon triangles (default 5 022 = diablo's polygon count),
    vertex normals = position / r, uv = (lon/2pi, lat/pi) clamped to [0.001, 0.999], and
    four deterministic textures (diffuse, two normal maps, specular exponent 0..64)."""
    lat = np.linspace(0.0, np.pi, n_lat + 1, dtype=np.float64)
    lon = np.linspace(0.0, 2.0 * np.pi, n_lon + 1, dtype=np.float64)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    nrm = np.stack([np.sin(la) * np.sin(lo), np.cos(la), np.sin(la) * np.cos(lo)], -1).reshape(-1, 3)
    pos = (nrm * radius).astype(np.float32)
    nrm = nrm.astype(np.float32)
    uv = np.stack([np.clip(lo / (2 * np.pi), 0.001, 0.999), np.clip(la / np.pi, 0.001, 0.999),
                   np.zeros_like(la)], -1).reshape(-1, 3).astype(np.float32)
    idx = []
    stride = n_lon + 1
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * stride + j, i * stride + j + 1
            c, d = (i + 1) * stride + j, (i + 1) * stride + j + 1
            idx.append([a, a, a, c, c, c, b, b, b])
            idx.append([b, b, b, c, c, c, d, d, d])
    mesh = {"pos": pos, "tex": uv, "nrm": nrm, "idx": np.array(idx, np.uint32)}

    diffuse = (_noise_image(0x5EED0000, tex_size) * 200.0 + 40.0).astype(np.uint8)

    def normal_map(seed):
        n = (_noise_image(seed, tex_size) - 0.5) * 0.4
        n[..., 2] = 1.0
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        return np.clip((n * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)

    spec = (_noise_image(0x5EED0003, tex_size)[..., :1] * 64.0).astype(np.uint8).repeat(3, axis=-1)
    return mesh, [diffuse, normal_map(0x5EED0001), normal_map(0x5EED0002), spec]


def instanced_grid(mesh, n=8):
    """BASELINE.json configs[4]: n x n grid of scaled copies, instance (i, j) =
    p / n + ((2i+1)/n - 1, (2j+1)/n - 1, 0), shared normals and uvs, polygons concatenated in
    (i major, j minor) instance order.  The reference has no instancing; this rule is ours."""
    pos, idx = mesh["pos"], mesh["idx"]
    n_pos = pos.shape[0]
    all_pos, all_idx = [], []
    for i in range(n):
        for j in range(n):
            off = np.array([(2 * i + 1) / n - 1.0, (2 * j + 1) / n - 1.0, 0.0], np.float32)
            all_pos.append((pos * np.float32(1.0 / n) + off).astype(np.float32))
            k = idx.copy()
            k[:, 0::3] += np.uint32((i * n + j) * n_pos)
            all_idx.append(k)
    return {"pos": np.concatenate(all_pos), "tex": mesh["tex"], "nrm": mesh["nrm"],
            "idx": np.concatenate(all_idx)}
