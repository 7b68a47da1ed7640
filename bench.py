#!/usr/bin/env python3
"""bench.py — frames/s (and shaded Mpix/s) of the Geometry Pass + Opaque Pass on a 4K Sponza-class scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one frame: camera write (the per-frame dirty upload of a static scene), geometry pass (deform/transform,
bin, raster) and the single-dispatch opaque pass, driven through the C++ host layer and the C-ABI.  Scene data is
resident in HBM before the timed region.  With N > 1 the frame is sharded into N horizontal strips (one process per
GPU); every step ends with an RCCL all-gather of the RGBA16F strips so that every rank holds the full image.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      for the dominant kernel: algorithmic bytes per launch / hipEvent-measured launch time vs 8 TB/s
  cpu_baseline  the scalar-C oracle port timed on this box's host cores on a bounded strip of the same frame
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(scene, stats, rows):
    """Compulsory HBM traffic per launch, SURVEY.md §8(d) (restated in DESIGN.md §"Measurement")."""
    import numpy as np
    W = scene.width
    P = W * rows
    P_cov = stats["covered_pixels"]
    T_in, T_bin, E = stats["triangles_in"], stats["triangles_binned"], stats["bin_entries"]
    V = 3 * T_in
    prims = [p for n in scene.nodes for p in n.primitives]
    stride = float(np.mean([8 * len(p.uvs) + 16 * len(p.colors) for p in prims])) if prims else 0.0
    tex_bytes = float(sum(t.nbytes for t in scene.textures))
    n_tex = float(np.mean([sum(1 for a in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "occlusion_tex", "emissive_tex")
                                if getattr(m, a) is not None) for m in scene.materials])) if scene.materials else 0.0
    u_tex = min(tex_bytes, P_cov * n_tex * 16.0)
    return {
        "k_deform_transform": V * 56.0 + V * 48.0 + T_in,
        "k_bin": 2.0 * (T_in * 49.0) + E * 4.0 * 2.0,
        "k_raster_tile": E * (4.0 + 48.0) + P * 8.0,
        "k_shade": P * 16.0 + P_cov * (12.0 + 3.0 * stride) + min(T_bin, P_cov) * 144.0 + u_tex,
    }


def pmc_traffic(kernel, n_tris, W, H):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/latest_pmc.json, written by
    tools/profile_collect.py from FETCH_SIZE / WRITE_SIZE collected in separate passes on this same workload).  None when
    no profile of this exact workload is committed: counters cannot be read from inside the process."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        doc = json.load(open(path))
        wl = doc["workload"]
        if (wl["triangles"], wl["width"], wl["height"]) != (n_tris, W, H):
            return None
        return doc["kernels"]["awsm::" + kernel]["hbm_traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(scene, lut_rg, rows_sample):
    """Oracle (oracle/c/*.c, scalar f32, -O2) on a bounded strip of the same frame, all host cores (row bands)."""
    from oracle import oracle_lib
    from tests import helpers
    threads = os.cpu_count() or 1
    model = helpers.build_model(scene)
    fr = oracle_lib.frame_from_model(model, lut_rg, rows=rows_sample)
    t0 = time.perf_counter()
    fr.transform()
    fr.raster(threads)
    fr.shade(threads)
    dt = time.perf_counter() - t0
    frac = (rows_sample[1] - rows_sample[0]) / scene.height
    return {"value": frac / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"rows [{rows_sample[0]},{rows_sample[1]}) of the {scene.width}x{scene.height} frame ({frac:.4f} frame, all vertices "
                      f"transformed), {dt:.1f} s wall, scaled to whole frames"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--detail", type=float, default=1.0, help="tessellation scale of the Sponza-class scene (1.0 = 262,144 triangles)")
    ap.add_argument("--tex-scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-frames", type=int, default=30)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # backend "nccl" IS RCCL on ROCm

    from awsm_renderer_amd import scenes
    from awsm_renderer_amd.host import Renderer
    from awsm_renderer_amd.sharding import gather_image, strip_rows

    W, H = args.width, args.height
    scene = scenes.atrium_scene(W, H, detail=args.detail, tex_scale=args.tex_scale)
    n_tris = scenes.total_triangles(scene)
    stream = torch.cuda.current_stream()
    r = Renderer(scene, device=local_rank, stream=stream.cuda_stream, lut_size=1024)
    y0, y1, per = strip_rows(H, world, rank)
    full = torch.zeros((world * per, W, 4), dtype=torch.float16, device="cuda")   # padded to equal strips for the all-gather
    if world > 1:
        strip = torch.zeros((per, W, 4), dtype=torch.float16, device="cuda")
        # the kernels address the image by absolute row: bind a base pointer such that row y0 lands on strip[0]
        r.host.set_shard_rows(y0, y1)
        base = strip.data_ptr() - y0 * W * 8
        from awsm_renderer_amd.hip_backend import HipDevice
        dev = HipDevice.from_ctx(r.host.device_ctx, W, H)
        dev.bind_output(base, H * W * 8)
    else:
        from awsm_renderer_amd.hip_backend import HipDevice
        dev = HipDevice.from_ctx(r.host.device_ctx, W, H)
        dev.bind_output(full.data_ptr(), H * W * 8)

    def step():
        r.host.camera_update(scene.view, scene.proj, scene.camera_position)
        r.host.render(sync=False)
        if world > 1:
            gather_image(strip, full, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first = r.render(sync=True)            # uploads everything, sizes the bin list
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    fps = args.steps / dt

    # ---- per-kernel launch durations: hipEvents recorded by the library on the kernels' own stream ----
    acc = {}
    for _ in range(max(1, args.profile_frames)):
        r.host.camera_update(scene.view, scene.proj, scene.camera_position)
        st = r.host.render(sync=True)
        for k, v in st.items():
            acc[k] = acc.get(k, 0.0) + float(v)
    st = {k: v / max(1, args.profile_frames) for k, v in acc.items()}
    kernel_ms = {"k_deform_transform": st["ms_transform"], "k_bin": st["ms_bin"], "k_raster_tile": st["ms_raster"], "k_shade": st["ms_shade"]}
    alg = algorithmic_bytes(scene, {k: int(round(v)) for k, v in st.items() if not k.startswith("ms_")}, y1 - y0)
    dom = max(kernel_ms, key=kernel_ms.get)
    achieved = alg[dom] / (kernel_ms[dom] * 1e-3) / 1e9 if kernel_ms[dom] > 0 else 0.0
    traffic = pmc_traffic(dom, n_tris, W, H) if world == 1 else None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": alg[dom], "launch_ms": kernel_ms[dom],
                "all_kernels_ms": kernel_ms, "all_kernels_algorithmic_bytes": alg}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_lib
        lut_rg = oracle_lib.brdf_lut(64, 64)
        mid = H // 2
        half = max(8, H // 8)
        cpu = cpu_baseline(scene, lut_rg, (max(0, mid - half), min(H, mid + half)))

    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "frames/sec + shaded Mpix/s, 4K Sponza glTF, 1/2/4/8 MI355X",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "shaded_mpix_per_s": W * H * fps / 1e6,
            "config": {"workload": f"Sponza-class procedural atrium (configs[3]): {n_tris} triangles, {len(scene.materials)} materials, "
                                   f"{len(scene.textures)} textures, {W}x{H}, geometry pass + opaque pass, single-sample, MipmapMode::None",
                       "triangles": n_tris, "width": W, "height": H,
                       "sharding": "none" if world == 1 else f"{world} horizontal strips of {per} rows + RCCL all-gather of the RGBA16F image",
                       "draws": len(r.host.draw_list())},
            "frame_stats": {k: st[k] for k in ("triangles_in", "triangles_binned", "bin_entries", "covered_pixels")},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    dev.close()
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
