"""Per-rank kernel times of a band-sharded 4K frame on ONE GPU (no collective): what each of N ranks would spend."""
import sys; sys.path.insert(0, '.')
from awsm_renderer_amd import scenes
from awsm_renderer_amd.hip_backend import HipDevice
from tests import helpers

W, H = 3840, 2160
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scenes.atrium_scene(W, H)
model = helpers.build_model(sc)
dev = HipDevice()
dev.resize(W, H, 0)
dev.upload_mirrors(model.mirrors())
for i, t in enumerate(model.texture_arrays()):
    dev.texture_array_upload(i, t["texels"])
for i, s in enumerate(sc.samplers):
    dev.sampler_set(i, s)
dev.env_upload(sc.skybox_rgba, sc.prefiltered_rgb, sc.irradiance_rgb)
dev.brdf_lut_generate(256, 256)
od = model.collect_draws()
draws = HipDevice.make_draws(od)
for n, ranks in ((1, [0]), (N, list(range(N)))):
    for r in ranks:
        dev.set_shard_bands(n, r, True) if n > 1 else dev.set_shard_bands(1, 0, False)
        acc = {}
        steps = 20
        for it in range(steps + 3):
            dev.geometry_pass(draws, len(od)); dev.opaque_pass(mipmap=0); st = dev.frame_end()
            if it >= 3:
                for k, v in st.items():
                    acc[k] = acc.get(k, 0) + v
        a = {k: round(v / steps, 4) for k, v in acc.items()}
        print(f"N={n} rank {r}: transform {a['ms_transform']:.3f} bin {a['ms_bin']:.3f} raster {a['ms_raster']:.3f} shade {a['ms_shade']:.3f} total {a['ms_total']:.3f}  binned {a['triangles_binned']:.0f} entries {a['bin_entries']:.0f}")
