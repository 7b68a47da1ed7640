"""The cube-sampling contract of the oracle (oracle/c/oracle_shade.c sample_cube: skybox.wgsl:37, brdf.wgsl:268-290) on the CPU:
face selection and orientation against the WebGPU / Vulkan table, seam-free filtering across edges, level blending."""
import numpy as np

from awsm_renderer_amd import scenes
from oracle import oracle_lib


def _cube_from(fn, n, levels=1):
    return [fn(scenes.cube_face_directions(max(n >> l, 1))).astype(np.float16) for l in range(levels)]


def test_face_table_and_texel_centres():
    n = 8
    faces = np.zeros((6, n, n, 4), dtype=np.float16)
    for f in range(6):
        faces[f, :, :, 0] = f                                   # r = face index
        faces[f, :, :, 1] = np.arange(n)[None, :]               # g = column
        faces[f, :, :, 2] = np.arange(n)[:, None]               # b = row
    d = scenes.cube_face_directions(n).reshape(-1, 3)
    got = oracle_lib.sample_cube([faces], d, np.zeros(len(d))).reshape(6, n, n, 4)
    assert np.abs(got[..., :3] - faces[..., :3].astype(np.float32)).max() < 1e-4      # a direction through a texel centre returns that texel
    axes = oracle_lib.sample_cube([faces], np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32) * 3.0, np.zeros(6))
    assert np.allclose(axes[:, 0], np.arange(6))                 # +X -X +Y -Y +Z -Z, any vector length


def test_constant_and_linear_fields_are_reproduced():
    n = 16
    const = _cube_from(lambda d: np.concatenate([np.full(d.shape[:-1] + (3,), 0.625), np.ones(d.shape[:-1] + (1,))], axis=-1), n, levels=5)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(5000, 3)).astype(np.float32)
    got = oracle_lib.sample_cube(const, d, rng.uniform(-1.0, 6.0, size=5000))
    assert np.allclose(got[:, :3], 0.625) and np.allclose(got[:, 3], 1.0)


def test_filtering_is_continuous_across_face_edges():
    """Two directions a hair apart on either side of an edge (and anywhere else away from the eight corners) must sample almost the same
    colour: taps beyond a face come from the adjacent face.  A clamp-to-face sampler fails this by half a texel's contrast."""
    n = 16
    rng = np.random.default_rng(5)
    noise = rng.uniform(0.0, 4.0, size=(6, n, n, 4)).astype(np.float16)        # uncorrelated texels: worst case for a seam
    worst = 0.0
    for a, b in ((0, 1), (0, 2), (1, 2)):
        for sa in (-1.0, 1.0):
            for sb in (-1.0, 1.0):
                t = rng.uniform(-0.85, 0.85, size=400)                        # along the edge, away from the corners
                d0 = np.zeros((400, 3)); d1 = np.zeros((400, 3))
                c = 3 - a - b
                d0[:, a] = sa; d0[:, b] = sb * (1.0 - 1e-4); d0[:, c] = t
                d1[:, a] = sa * (1.0 - 1e-4); d1[:, b] = sb; d1[:, c] = t
                g0 = oracle_lib.sample_cube([noise], d0.astype(np.float32), np.zeros(400))
                g1 = oracle_lib.sample_cube([noise], d1.astype(np.float32), np.zeros(400))
                worst = max(worst, float(np.abs(g0 - g1).max()))
    assert worst < 4.0 * n * 2e-4 * 2, worst       # |d colour| <= contrast * texels-per-unit * step


def test_levels_blend_linearly_and_clamp():
    n = 8
    lv = [np.full((6, max(n >> l, 1), max(n >> l, 1), 4), float(l), dtype=np.float16) for l in range(4)]
    d = np.tile(np.array([[0.2, -0.3, 0.9]], dtype=np.float32), (6, 1))
    lods = np.array([-2.0, 0.0, 0.25, 1.5, 3.0, 9.0], dtype=np.float32)
    got = oracle_lib.sample_cube(lv, d, lods)[:, 0]
    assert np.allclose(got, [0.0, 0.0, 0.25, 1.5, 3.0, 3.0])


def test_procedural_environment_shapes():
    env = scenes.procedural_environment(32, 8)
    assert [a.shape for a in env["prefiltered"]] == [(6, max(32 >> l, 1), max(32 >> l, 1), 4) for l in range(6)]
    assert env["skybox"][0].dtype == np.float16 and float(env["skybox"][0][..., :3].max()) > 4.0      # HDR
