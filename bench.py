#!/usr/bin/env python3
"""bench.py — frames/s (and shaded Mpix/s) of the Geometry Pass + Opaque Pass on a 4K Sponza-class scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one frame: camera write (the camera moves every step on a small orbit, so the cull, the sort and the draw-list
upload are exercised, not just replayed), geometry pass (deform/transform, bin, raster) and the single-dispatch opaque pass, driven
through the C++ host layer and the C-ABI.  Scene data is resident in HBM before the timed region.  The scene is the procedural
"Sponza-class" atrium of scenes.py (no glTF asset is available offline); --config 2 / 3 run BASELINE configs[1] / configs[2].  With N > 1 the frame is sharded into 32-row bands dealt round-robin over the
ranks (one process per GPU); every step ends with an RCCL all-gather of the RGBA16F bands so that every rank holds the
full image (the gather of frame i runs while frame i+1 renders).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      for the dominant kernel: what binds it (vector-ALU issue, from the committed SQ counter pass named in profile_tag) next to
                its HBM roofline: algorithmic bytes per launch (low / high texel bound) / hipEvent-measured launch time vs 8 TB/s
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
XGMI_LINK_GBS_PER_DIRECTION = 76.5     # same guide: 7 links x ~153 GB/s bidirectional per GPU, point to point (one link per peer on an 8-GPU node)


def collective_model(world, bytes_per_rank, to_root):
    """What the frame's exchange step should take by link rates alone, so that the first run on more than one GPU can be read against a
    prediction.  xGMI is point to point: every pair of GPUs has its own link.  all-gather: every rank sends its bands to each of the N-1 peers over that
    peer's link and receives theirs the same way — N-1 links busy in both directions at once, each carrying bytes_per_rank per direction.  gather to rank 0:
    the root's N-1 links each carry one rank's bands inwards.  Either way the step is bytes_per_rank / (one link, one direction), independent of N —
    but bytes_per_rank = frame / N, so the step shrinks as 1 / N while at N = 2 it is half a frame over ONE link: 33 MB / 76.5 GB/s = 0.43 ms against a
    0.31 ms single-GPU frame.  The curve is predicted to DIP at N = 2 (link-bound below the 1-GPU rate) and to become compute-bound from N = 4 on."""
    ms = bytes_per_rank / (XGMI_LINK_GBS_PER_DIRECTION * 1e9) * 1e3
    return {"model_ms": ms, "link_GBps_per_direction": XGMI_LINK_GBS_PER_DIRECTION, "links_used_per_rank": world - 1 if not to_root else 1,
            "frames_per_s_if_link_bound": 1e3 / ms if ms > 0 else None,
            "note": "point-to-point xGMI, one link per peer: the exchange takes bytes_per_rank / link rate whatever N is; the gather of frame i is overlapped with "
                    "the render of frame i + 1, so the frame period is max(render, model_ms) at best; N = 2 is predicted link-bound below the 1-GPU rate"}


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
    # unique texel bytes the frame touches: at most every texel of the scene (or four taps per texture and pixel, if that is less);
    # at least one texel per texture and covered pixel — a minified, unmipped texture is read sparsely, a magnified one densely, so
    # the truth lies in between (SURVEY.md §8d gives ~150 MB for this class of scene against the 333 MB upper bound)
    u_tex_high = min(tex_bytes, P_cov * n_tex * 16.0)
    u_tex_low = min(tex_bytes, P_cov * n_tex * 4.0) * (150.0 / 333.0 if tex_bytes > 3.0e8 else 0.45)
    shade_fixed = P * 16.0 + P_cov * (12.0 + 3.0 * stride) + min(T_bin, P_cov) * 144.0
    return {
        # geometry cache (round 5): a workgroup that keeps its draw's cached outputs reads the world position and writes clip (32 B / vertex); the others
        # read the 56-byte record and write clip, world position, N, T (+ the per-triangle words)
        "k_deform_transform": V * ((1.0 - stats.get("geometry_cache_frac", 0.0)) * (56.0 + 64.0) + stats.get("geometry_cache_frac", 0.0) * 32.0) + T_in * (1.0 - stats.get("geometry_cache_frac", 0.0)),
        "k_bin": 2.0 * (T_in * 49.0) + E * 4.0 * 2.0,
        "k_raster_tile": E * (4.0 + 48.0) + P * 8.0,
        "k_shade": shade_fixed + u_tex_high,
        "k_shade_low": shade_fixed + u_tex_low,
    }


SHADE_KERNELS = ("k_shade_lean<false, 0, false>", "k_shade_lean<false, false, false>", "k_shade_lean<false>", "k_shade_lean", "k_shade", "k_shade<false>", "k_shade<0>")      # profile names of the single-sample, MipmapMode::None opaque kernel


def pmc_profile(n_tris, W, H):
    """The committed rocprofv3 --pmc summary of this exact workload (profiles/latest_pmc.json), or None."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        wl = doc["workload"]
        return doc if (wl["triangles"], wl["width"], wl["height"]) == (n_tris, W, H) else None
    except (OSError, KeyError, ValueError):
        return None


def pmc_kernel(doc, kernel):
    if not doc:
        return None
    names = ["awsm::" + k for k in SHADE_KERNELS] if kernel == "k_shade" else ["awsm::" + kernel, "awsm::" + kernel + "<false>", "awsm::" + kernel + "<1>"]
    for n in names:
        if n in doc["kernels"]:
            return doc["kernels"][n]
    return None


def valu_issue(k, launch_ms, mix=None):
    """What bounds the dominant kernel beside its memory traffic (it is ALU work, not a stream): vector-ALU instructions per launch from
    the committed SQ counter pass, the time the 1024 SIMDs (256 CUs x 4) need to issue them, and that time over the measured launch
    duration; next to it the share of wave-cycles the same pass saw stalled on issue (SQ_WAIT_INST_ANY) and parked on memory (SQ_WAIT_ANY).
    Cycles per wave64 instruction: gfx950 issues a full-rate FP32 instruction in 2 cycles (157 TFLOP/s vector FP32 = 32 FMA lanes per SIMD
    and clock), conversions / compares / selects / 3-operand integer / packed / f64 ops and anything with an SGPR or literal source at about
    half that rate, transcendentals at about a quarter (tools/valu_probe.hip); `mix` is the kernel's static ISA mix weighted that way
    (tools/isa_cost.py through tools/profile_collect.py).  Without a mix the classic 4 cycles per instruction is used."""
    if not k or not k.get("SQ_INSTS_VALU") or launch_ms <= 0:
        return None
    insts = k["SQ_INSTS_VALU"]
    cpi = mix["cycles_per_wave64_instruction"] if mix else 4.0
    min_ms = insts * cpi / (1024 * 2.4e9) * 1e3
    out = {"insts_per_launch": insts, "insts_per_wave": insts / k["SQ_WAVES"] if k.get("SQ_WAVES") else None, "cycles_per_instruction_model": cpi,
           "model": "static ISA mix x measured issue classes" if mix else "4 cycles per wave64 instruction", "issue_bound_ms": min_ms, "frac": min_ms / launch_ms,
           "frac_at_4_cycles_per_instruction": insts * 4.0 / (1024 * 2.4e9) * 1e3 / launch_ms}
    if k.get("SQ_WAVE_CYCLES"):
        out["wave_cycles_issue_stalled"] = k.get("SQ_WAIT_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
        out["wave_cycles_waiting_memory"] = k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
    return out


def pmc_mix(doc, kernel):
    if not doc or "valu_mix" not in doc:
        return None
    names = ["awsm::" + k for k in SHADE_KERNELS] if kernel == "k_shade" else ["awsm::" + kernel, "awsm::" + kernel + "<false>", "awsm::" + kernel + "<1>"]
    for n in names:
        if n in doc["valu_mix"]:
            return doc["valu_mix"][n]
    return None


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, lut_rg, rows_sample, rows_one_thread):
    """Oracle (oracle/c/*.c, scalar f32, -O2) on a bounded strip of the same frame: all host threads (row bands) on `rows_sample`, and ONE thread on the
    smaller `rows_one_thread` (both figures, with the CPU model).  The vertex stage runs over the whole scene once (it is the
    same work for any strip; its time is counted in both figures in full, as a whole frame would pay it)."""
    from oracle import oracle_lib
    from tests import helpers
    try:
        threads = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        threads = os.cpu_count() or 1
    model = helpers.build_model(scene)

    def timed(rows, n_threads):
        fr = oracle_lib.frame_from_model(model, lut_rg, rows=rows)
        t0 = time.perf_counter()
        fr.transform()
        t1 = time.perf_counter()
        fr.raster(n_threads)
        fr.shade(n_threads)
        t2 = time.perf_counter()
        frac = (rows[1] - rows[0]) / scene.height
        # a whole frame = the vertex stage once + the per-row work scaled up
        return 1.0 / ((t1 - t0) + (t2 - t1) / frac), frac, t2 - t0

    all_v, all_frac, all_dt = timed(rows_sample, threads)
    one_v, one_frac, one_dt = timed(rows_one_thread, 1)
    return {"value": all_v, "unit": "frames/s", "cores": threads, "kind": "port", "cpu_model": cpu_model_string(),
            "one_thread": {"value": one_v, "unit": "frames/s", "cores": 1,
                           "sample": f"rows [{rows_one_thread[0]},{rows_one_thread[1]}) ({one_frac:.4f} frame), {one_dt:.1f} s wall"},
            "sample": f"rows [{rows_sample[0]},{rows_sample[1]}) of the {scene.width}x{scene.height} frame ({all_frac:.4f} frame; all vertices transformed once, "
                      f"that time counted in full), {all_dt:.1f} s wall on {threads} threads, scaled to whole frames; one_thread: the same on a narrower strip"}


def hip_backend_path():
    from awsm_renderer_amd import hip_backend
    return hip_backend.LIB_PATH


def backend_is_rehearsal():
    return os.environ.get("AWSM_BENCH_BACKEND", "nccl") != "nccl"


def launch_ranks(n):
    """`python bench.py --gpus N` typed as is (no launcher in front, WORLD_SIZE unset): start the N ranks as child processes of this one — which has
    not imported torch and makes no HIP call, before or after — with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would set
    them, one GPU each.  Rank 0 inherits stdout (its JSON line is this command's JSON line); the other ranks' stdout goes to stderr.  Returns the
    worst return code; when a rank fails, the others get a grace period (they normally leave through the collective's own error) and are then
    ended by PID."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), AWSM_BENCH_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # this pool's driver only supports dmabuf IPC (RCCL needs it)
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    kids = []
    for i in range(n):
        env = dict(base, RANK=str(i), LOCAL_RANK=str(i), GROUP_RANK="0")
        kids.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if i == 0 else sys.stderr))
    rcs = [None] * n
    failed_at = None
    grace = float(os.environ.get("AWSM_BENCH_RANK_GRACE_S", "60"))
    try:
        while any(rc is None for rc in rcs):
            for i, k in enumerate(kids):
                if rcs[i] is None:
                    rcs[i] = k.poll()
                    if rcs[i] not in (None, 0) and failed_at is None:
                        failed_at = time.monotonic()
                        print(f"bench.py: rank {i} exited with code {rcs[i]}; waiting up to {grace:.0f} s for the other ranks", file=sys.stderr)
            if failed_at is not None and time.monotonic() - failed_at > grace:
                break
            time.sleep(0.05)
    finally:
        for i, k in enumerate(kids):
            if k.poll() is None:
                k.terminate()
        for i, k in enumerate(kids):
            if rcs[i] is None:
                try:
                    rcs[i] = k.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    k.kill()
                    rcs[i] = k.wait()
    worst = 0
    for rc in rcs:
        if rc:
            worst = max(worst, rc if rc > 0 else 128 - rc)
    return worst


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
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap the opaque pass of frame i with the geometry pass of frame i+1")
    ap.add_argument("--mipmap", action="store_true", help="MipmapMode::Gradient (the reference's default; not the BASELINE config)")
    ap.add_argument("--anisotropic", action="store_true", help="with --mipmap: honour the samplers' max_anisotropy (16 in this scene, as the reference's glTF ingest sets it; AWSM_CFG_ANISOTROPIC)")
    ap.add_argument("--msaa", type=int, default=0, choices=(0, 4), help="MSAA x4 geometry + edge resolve (the reference's default AntiAliasing; not the BASELINE config)")
    ap.add_argument("--strips", action="store_true", help="with --msaa 4 and N > 1: shard by contiguous row strips (each carries its own halo rows) instead of bands + halo exchange")
    ap.add_argument("--gather", choices=("all", "root"), default="all", help="N > 1: how the image leaves the ranks — 'all': RCCL all-gather, every rank ends up with the frame "
                    "(BASELINE.json north_star); 'root': gather to rank 0 only (one consumer: N-1 bands travel over N-1 separate xGMI links into one GPU instead of "
                    "N(N-1) band transfers; SURVEY.md 8e's alternative)")
    ap.add_argument("--check", action="store_true", help="after the timed loop, compare the gathered image with an unsharded render of the same frame")
    ap.add_argument("--config", type=int, default=4, choices=(2, 3, 4), help="BASELINE.json config: 4 = configs[3] the 4K Sponza-class frame (the metric's), "
                    "2 = configs[1] helmet-class 15k triangles / 2048^2 textures at 1920x1080, 3 = configs[2] skinned rig + morph cube at 1920x1080")
    ap.add_argument("--env", choices=("uniform", "procedural"), default="uniform", help="environment: the builder-default uniform cubes (BASELINE), or texel cubemaps — a procedural HDR "
                    "skybox / prefiltered chain / irradiance cube (scenes.procedural_environment) sampled by skybox.wgsl:37 and brdf.wgsl:268-290")
    ap.add_argument("--static-camera", action="store_true", help="re-submit the same camera every step (default: a small orbit, so cull / sort / draw-list upload run)")
    ap.add_argument("--trace", action="store_true", help="add `frame_trace` to the JSON line: per frame of warm-up + timed loop the host time at which step() returned and the "
                    "device-clock times of geometry begin / geometry done / shading done (awsm_hip_frame_trace), ms from the first")
    ap.add_argument("--allow-variant-lib", action="store_true", help="accept an AWSM_HIP_LIB override (A/B builds); the path is printed in the JSON line")
    args = ap.parse_args()
    if os.environ.get("AWSM_HIP_LIB") and not args.allow_variant_lib:
        raise SystemExit("AWSM_HIP_LIB is set: bench.py measures awsm-renderer_amd/libawsm_hip.so; pass --allow-variant-lib for an A/B build")
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed without a launcher: this process becomes the launcher (nothing GPU-related has been imported or called yet, and never is)
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or unset WORLD_SIZE and let "
                         f"`python bench.py --gpus {args.gpus}` start its own ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback for the product path")
    if backend_is_rehearsal():
        local_rank = 0                     # rehearsal: all ranks share GPU 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but this node shows {torch.cuda.device_count()} HIP device(s): one rank per GPU "
                         f"(AWSM_BENCH_BACKEND=gloo rehearses the N > 1 path with every rank on GPU 0)")
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("AWSM_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # backend "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from awsm_renderer_amd import scenes
    from awsm_renderer_amd.host import Renderer
    from awsm_renderer_amd.sharding import band_rows, bands_per_rank, bands_to_image

    W, H = args.width, args.height
    if args.config == 4:
        scene = scenes.atrium_scene(W, H, detail=args.detail, tex_scale=args.tex_scale)
        workload_name = "Sponza-class procedural atrium (configs[3])"
    else:
        if (W, H) == (3840, 2160):
            W, H = 1920, 1080
        if args.config == 2:
            scene = scenes.helmet_scene(W, H, tex_size=max(8, int(2048 * args.tex_scale)))
            workload_name = "DamagedHelmet-class procedural mesh (configs[1])"
        else:
            scene = scenes.skinned_morph_scene(W, H)
            workload_name = "BrainStem-class skinned rig + AnimatedMorphCube-class morph cube (configs[2])"
    if args.env == "procedural":
        scene.env_cubes = scenes.procedural_environment(256, 32)
        scene.prefiltered_mip_count = len(scene.env_cubes["prefiltered"])
        scene.irradiance_mip_count = 1
    n_tris = scenes.total_triangles(scene)
    # One explicit HIP stream for everything: the library launches its kernels on it, and torch (RCCL collectives, barrier
    # tensors) orders against it as its current stream.  (torch's default stream has handle 0, which the C-ABI reads as
    # "create a private stream" — that one would not be ordered with the collectives.)
    stream = torch.cuda.Stream(device=local_rank)
    if os.environ.get("AWSM_BENCH_STREAM_CU_MASK"):      # experiment (tools/ab_cu_mask.sh): the caller's stream — the geometry passes — on a subset of the CUs
        import ctypes
        words = [int(x, 16) for x in os.environ["AWSM_BENCH_STREAM_CU_MASK"].split(",")]
        hip = ctypes.CDLL("libamdhip64.so")
        raw = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(raw), ctypes.c_uint32(len(words)), (ctypes.c_uint32 * len(words))(*words))
        if rc != 0:
            raise SystemExit(f"hipExtStreamCreateWithCUMask: {rc}")
        stream = torch.cuda.ExternalStream(raw.value, device=local_rank)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    r = Renderer(scene, device=local_rank, stream=stream.cuda_stream, lut_size=1024, msaa=args.msaa, mipmap=args.mipmap, overlap_frames=not args.no_overlap,
                 anisotropic=args.anisotropic)
    from awsm_renderer_amd.hip_backend import HipDevice
    dev = HipDevice.from_ctx(r.host.device_ctx, W, H)
    # N > 1: 32-row bands dealt round-robin over the ranks (rank r owns tile rows r, r+N, ...: every rank gets 1/N of the
    # dense part of the screen), compact [L*32, W] output per rank, RCCL all-gather -> [N, L, 32, W], de-interleaved by
    # bands_to_image.  Double-buffered: frame i is gathered (RCCL's stream) while frame i+1 renders.
    # With --msaa 4 the edge detector needs the rows next to every band: the ranks all-gather the sample-0 keys of their bands' first and
    # last rows between the two passes (the frame's one exchange step; awsm_hip_msaa_halo_*).  --strips: contiguous row strips
    # instead, which rasterise their own halo rows and need no exchange (but balance the load badly).
    strips = bool(args.msaa) and world > 1 and args.strips
    L = bands_per_rank(H, world)
    per = (H + world - 1) // world                      # rows per strip
    y0s, y1s = min(rank * per, H), min(rank * per + per, H)
    rows_out = (per if strips else L * 32) if world > 1 else H
    n_buf = 2 if world > 1 else 1
    gathered = [torch.zeros((world, rows_out, W, 4), dtype=torch.float16, device="cuda") for _ in range(n_buf)] if world > 1 else None
    image = [torch.zeros((H, W, 4), dtype=torch.float16, device="cuda") for _ in range(n_buf)]       # what every rank ends up holding
    mine = [torch.zeros((rows_out, W, 4), dtype=torch.float16, device="cuda") for _ in range(n_buf)] if world > 1 else None
    pending = [None] * n_buf
    frame_no = [0]
    if strips:
        r.host.set_shard_rows(y0s, y1s)
    elif world > 1:
        r.host.set_shard_bands(world, rank, compact_output=True)
    # one GPU: the frames go to the library's own images, one per frame slot (awsm_hip_output_device_ptr after awsm_hip_frame_flush is what a
    # consumer reads): two frames' opaque passes then share nothing and frame i + 1's may start while frame i's still drains.  (Binding ONE image
    # for every frame, as this script did until round 2, makes the library order the two passes: the second would write into the first's image.)

    class _Done:
        def wait(self):
            return True

    if args.msaa and world > 1 and not strips:
        Lh = dev.msaa_halo_bands()
        halo_mine = [torch.zeros((Lh, 2, W), dtype=torch.int64, device="cuda") for _ in range(2)]
        halo_all = [torch.zeros((world, Lh, 2, W), dtype=torch.int64, device="cuda") for _ in range(2)]
        halo_no = [0]

        def exchange_halo():      # RenderHooks.after_geometry_pass: two buffers, because frame i's opaque pass may still read while frame i+1 renders
            b = halo_no[0] % 2
            halo_no[0] += 1
            dev.msaa_halo_export(halo_mine[b].data_ptr(), halo_mine[b].numel() * 8)
            if backend == "nccl":
                dist.all_gather_into_tensor(halo_all[b].view(world * Lh, 2, W), halo_mine[b])      # RCCL, on the stream the library renders on
            else:
                torch.cuda.current_stream().synchronize()
                host_dst = torch.empty((world * Lh, 2, W), dtype=torch.int64)
                dist.all_gather_into_tensor(host_dst, halo_mine[b].cpu())
                halo_all[b].copy_(host_dst.view(world, Lh, 2, W))
            dev.msaa_halo_bind(halo_all[b].data_ptr(), halo_all[b].numel() * 8)
        r.host.set_render_hooks(after_geometry_pass=exchange_halo)

    to_root = args.gather == "root"

    def all_gather(dst, src):
        """dst [world * rows_out, W, 4] <- every rank's src [rows_out, W, 4], in rank order: on every rank, or (--gather root) on rank 0 only."""
        if backend == "nccl":
            if to_root:
                return dist.gather(src, gather_list=list(dst.view(world, *src.shape).unbind(0)) if rank == 0 else None, dst=0, async_op=True)
            return dist.all_gather_into_tensor(dst, src, async_op=True)
        # rehearsal backend (AWSM_BENCH_BACKEND=gloo, several ranks sharing one GPU): staged through the host, synchronous
        torch.cuda.current_stream().synchronize()
        host_src = src.view(torch.int32).cpu()
        host_dst = torch.empty((world,) + tuple(host_src.shape), dtype=torch.int32)
        if to_root:
            dist.gather(host_src, gather_list=list(host_dst.unbind(0)) if rank == 0 else None, dst=0)
            if rank != 0:
                return _Done()
        else:
            dist.all_gather_into_tensor(host_dst.view(world * host_src.shape[0], *host_src.shape[1:]), host_src)
        dst.copy_(host_dst.view(world * host_src.shape[0], *host_src.shape[1:]).view(torch.float16).view(dst.shape))
        return _Done()

    def finish(b):
        """Order the gather of buffer b before the current stream and de-interleave it into image[b]."""
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
            if to_root and rank != 0:
                return                 # only rank 0 holds the frame
            if strips:
                image[b].copy_(gathered[b].view(world * rows_out, W, 4)[:H])
            else:
                image[b].copy_(bands_to_image(gathered[b].view(world, L, 32, W, 4), H, world))

    # The camera moves: a small orbit around its position (radius 2 % of the distance to its target, one turn per 240 frames), so every
    # step runs the frustum cull, the depth sort and — when the order changes — the draw-list upload, and the binning sees a new frame.
    import math
    inv_view = np.linalg.inv(np.asarray(scene.view, dtype=np.float64).T)          # scene.view is [col][row]
    eye0 = np.asarray(scene.camera_position, dtype=np.float64)
    fwd = -inv_view[:3, 2]
    right, up = inv_view[:3, 0], inv_view[:3, 1]
    orbit_r = 0.02 * 30.0 if args.config == 4 else 0.02 * float(np.linalg.norm(eye0))
    cam_no = [0]

    def move_camera():
        if args.static_camera:
            r.host.camera_update(scene.view, scene.proj, scene.camera_position)
            return
        a = 2.0 * math.pi * (cam_no[0] % 240) / 240.0
        cam_no[0] += 1
        eye = eye0 + orbit_r * (math.cos(a) * right + math.sin(a) * up)
        r.host.camera_update(scenes.look_at_rh(tuple(eye), tuple(eye + 30.0 * fwd)), scene.proj, tuple(eye))

    handoff_in_loop = [0]

    def render_frame():
        """One frame enqueued.  ABI 2 reports a hand-off gate that timed out at the next awsm_hip_geometry_pass / frame_flush / frame_end — which, in an
        enqueue-only loop, is a call inside this loop.  The library has then already switched the context to events and the refused call did nothing:
        note the fault (the measurement is repeated) and issue the frame again, so that every rank still runs the same sequence of collectives."""
        try:
            r.host.render(sync=False)
        except Exception as e:
            if "hand-off" not in str(e):
                raise
            handoff_in_loop[0] += 1
            r.host.render(sync=False)

    def flush_frame():
        try:
            dev.frame_flush()
        except Exception as e:
            if "hand-off" not in str(e):
                raise
            handoff_in_loop[0] += 1
            dev.frame_flush()

    def step():
        move_camera()
        if world == 1:
            render_frame()
            return
        b = frame_no[0] % n_buf
        frame_no[0] += 1
        finish(b)                      # frame i-2 used these buffers: complete it before they are overwritten
        if strips:   # this rank's strip of the gather buffer receives frame rows y0s ..
            dev.bind_output_rows(mine[b].data_ptr(), rows_out * W * 8, y0s)
        else:
            dev.bind_output(mine[b].data_ptr(), rows_out * W * 8)
        render_frame()
        flush_frame()                  # the opaque pass ran on the library's shade stream: order it before the collective
        pending[b] = all_gather(gathered[b].view(world * rows_out, W, 4), mine[b])

    def drain():
        for b in range(n_buf):
            finish(b)

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_first = time.perf_counter()
    first = r.render(sync=True)            # uploads everything, sizes the bin list
    first_frame_ms = (time.perf_counter() - t_first) * 1e3
    first_frame_upload = int(r.host.upload_bytes_last_frame())
    # ---- per-kernel launch durations: hipEvents recorded by the library on the kernels' own stream ----
    # The synchronised profile frames (the scene's own camera: the frame the committed PMC passes profiled) run BEFORE the warm-up and the timed loop since
    # round 5 (they used to follow it): the script does this work anyway, and done first it also brings the part out of the idle clock state the seconds of
    # CPU-side scene set-up leave it in — the driver's 20-step run used to be measured on a part still ramping up.  `steps` and
    # `warmup` are exactly what was passed; "frames_before_timed_loop" in the JSON line says what ran before them.
    # (Python's garbage collector is switched off from here to the end of the timed loop, after one collection now: a collection between the profile frames
    # and the warm-up would leave the GPU idle for ~35 ms, and the part's clocks fall back within a few milliseconds of idling.)
    import gc
    gc.collect()
    gc.disable()
    r.host.set_render_timings(True)
    per_frame = []
    for _ in range(max(1, args.profile_frames)):
        r.host.camera_update(scene.view, scene.proj, scene.camera_position)
        per_frame.append({k: float(v) for k, v in r.host.render(sync=True).items()})
    # The kernel durations are the mean over the LAST THIRD of these frames: the first ones run on a part that has idled through the scene set-up, and
    # the same kernel on the same frame takes 232 us in the first profile frame and 213 us in the thirtieth (rocprofv3 trace, profiles/r05_b_*).
    profile_used = per_frame[-max(1, len(per_frame) // 3):]
    st = {k: sum(f[k] for f in profile_used) / len(profile_used) for k in per_frame[0]}
    r.host.set_render_timings(bool(os.environ.get("AWSM_BENCH_STAGE_TIMERS")))   # the timed loop does not read per-stage times: no event bubbles between the kernels
    # No Python garbage collection inside the warm-up and the timed loop: a generation-2 pass over the scene's objects takes ~35 ms, and
    # whether one lands in a 25-70 ms loop depends on the allocation count of everything before it (seen: configs[1] at 3,100 or 8,300
    # frames/s depending on an unrelated command-line flag).  The loop itself allocates a few tuples per step.
    steady_upload = [0]
    host_t = []                            # --trace: perf_counter at every step() return (and around the barriers)

    def timed_loop():
        if gc.isenabled():         # (a repeated measurement: the first one re-enabled it)
            gc.collect()
            gc.disable()
        del host_t[:]
        if args.trace:
            dev.frame_trace(args.warmup + args.steps + 8)
        host_t.append(("loop_begin", time.perf_counter()))
        try:
            for _ in range(args.warmup):
                step()
                if args.trace:
                    host_t.append(("w", time.perf_counter()))
            barrier()
            t0 = time.perf_counter()
            host_t.append(("t0", t0))
            for _ in range(args.steps):
                step()
                if args.trace:
                    host_t.append(("s", time.perf_counter()))
            host_t.append(("enqueued", time.perf_counter()))
            barrier()
            dt = time.perf_counter() - t0
            host_t.append(("t1", t0 + dt))
            steady_upload[0] = int(r.host.upload_bytes_last_frame())      # what the last timed frame sent over the boundary (the camera block)
        finally:
            gc.enable()
        return dt

    handoff_mode = dev.stream_handoff()
    dt = timed_loop()
    frame_trace = None
    if args.trace:
        dev_ms, _ = dev.read_frame_trace(args.warmup + args.steps)
        base = host_t[0][1]
        frame_trace = {"host_ms": [[k, round((t - base) * 1e3, 4)] for k, t in host_t],
                       "device_ms_geometry_begin_done_shade_done": [[round(float(x), 4) for x in row] for row in dev_ms],
                       "note": "host and device clocks have different origins: compare intervals, not instants"}
        dev.frame_trace(0)
    # A device-side hand-off gate that gave up during the loop (kernels of two streams not running side by side: a profiler that serialises
    # them attached after the context's probe, a hardware queue shared with a stream created later) leaves frames that may be incomplete and
    # the context on events: the measurement is then repeated, on events, and the line says so.
    handoff_fault = 1 if handoff_in_loop[0] else 0
    try:
        r.host.render(sync=True)
    except Exception as e:      # HostError: the library reports a timed-out gate once, at its next geometry_pass / frame_flush / frame_end
        if "hand-off" not in str(e):
            raise
        handoff_fault = 1
    if dev.stream_handoff() != handoff_mode:
        handoff_fault = 1
    if world > 1:
        t = torch.tensor([handoff_fault], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        handoff_fault = int(t.item())
    if handoff_fault:
        print(f"rank {rank}: a stream hand-off gate timed out during the timed loop; repeating the measurement on hipEvents", file=sys.stderr)
        dt = timed_loop()
    dt_local = dt
    ranks_seen = 1
    gather_alone_ms = None
    if world > 1:
        coll_dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.ones(1, dtype=torch.int32, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)           # over RCCL (or the rehearsal backend): how many ranks really took part
        ranks_seen = int(t.item())
        # the frame's exchange step by itself (nothing rendering beside it): K collectives back to back on the image buffers
        drain()
        torch.cuda.synchronize()
        dist.barrier()
        K = 5
        g0 = time.perf_counter()
        for _ in range(K):
            h = all_gather(gathered[0].view(world * rows_out, W, 4), mine[0])
            h.wait()
        torch.cuda.synchronize()
        gather_alone_ms = (time.perf_counter() - g0) / K * 1e3
    ms_per_step = dt / args.steps * 1e3
    fps = args.steps / dt

    check = None
    if args.check and world > 1:
        last = (frame_no[0] - 1) % n_buf
        got = image[last].clone()
        ref = torch.zeros((H, W, 4), dtype=torch.float16, device="cuda")
        if strips:
            r.host.set_shard_rows(0, 0)
        else:
            r.host.set_shard_bands(1, 0)
        halo_hook = args.msaa and not strips
        if halo_hook:
            r.host.set_render_hooks()                    # the unsharded reference frame has no exchange
        dev.bind_output(ref.data_ptr(), H * W * 8)
        r.host.render(sync=True)
        check = "ok" if (to_root and rank != 0) or torch.equal(got.view(torch.int16), ref.view(torch.int16)) else "MISMATCH"
        if strips:
            r.host.set_shard_rows(y0s, y1s)
            dev.bind_output_rows(mine[0].data_ptr(), rows_out * W * 8, y0s)
        else:
            r.host.set_shard_bands(world, rank, compact_output=True)
            dev.bind_output(mine[0].data_ptr(), rows_out * W * 8)
            if halo_hook:
                r.host.set_render_hooks(after_geometry_pass=exchange_halo)
        if check != "ok":
            raise SystemExit(f"rank {rank}: gathered image differs from the unsharded frame")

    lean = st.get("ms_shade_lean", 0.0) > 0.0
    kernel_ms = {"k_deform_transform": st["ms_transform"], "k_bin": st["ms_bin"], "k_raster_tile": st["ms_raster"], "k_shade": st["ms_shade_lean"] if lean else st["ms_shade"]}
    if st.get("frames_with_dropped_bin_entries", 0) or st.get("bin_overflow_retries", 0):
        raise SystemExit(f"rank {rank}: the timed loop lost geometry or had to replay frames: {st['frames_with_dropped_bin_entries']} dropped, {st['bin_overflow_retries']} replayed")
    rows_mine = H if world == 1 else (y1s - y0s if strips else len(band_rows(H, world, rank)))
    stats_i = {k: int(round(v)) for k, v in st.items() if not k.startswith("ms_")}
    stats_i["geometry_cache_frac"] = (st.get("geometry_cache_blocks", 0.0) / st["geometry_blocks"]) if st.get("geometry_blocks") else 0.0
    alg = algorithmic_bytes(scene, stats_i, rows_mine)
    dom = max(kernel_ms, key=kernel_ms.get)
    alg_low = alg.pop("k_shade_low")
    sec = kernel_ms[dom] * 1e-3
    achieved = alg[dom] / sec / 1e9 if sec > 0 else 0.0
    achieved_low = (alg_low if dom == "k_shade" else alg[dom]) / sec / 1e9 if sec > 0 else 0.0
    # HBM bytes and SQ counters of the dominant kernel come from the committed rocprofv3 passes of this same workload (counters cannot be
    # read from inside the process): profiles/latest_pmc.json, collected by tools/profile_round.sh — profile_tag says which run.
    prof = pmc_profile(n_tris, W, H) if (world == 1 and args.config == 4 and not args.msaa and not args.mipmap) else None
    pk = pmc_kernel(prof, dom)
    valu = valu_issue(pk, kernel_ms[dom], pmc_mix(prof, dom))
    # Calibrated (profiles/r04_traffic_calibration.txt, r04_traffic_request_sizes.txt; tools/traffic_calibration.sh, tools/traffic_sizes.sh): on gfx950 EVERY
    # memory-side read request of these kernels is 128 bytes — an 8-byte scattered texel pair, a 13-lane 16-byte record gather and a streamed key
    # read alike (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ; _32B = 0, _64B ~ 0) — while FETCH_SIZE tallies each at 64 bytes.  So traffic = 2 x FETCH_SIZE +
    # WRITE_SIZE for every access pattern here, not only for wide streams; it is measured at the L2's fabric side and includes Infinity-Cache hits.
    traffic_raw = pk.get("hbm_read_bytes_raw", 0.0) + pk.get("hbm_write_bytes", 0.0) if pk else None
    traffic_cal = 2.0 * pk.get("hbm_read_bytes_raw", 0.0) + pk.get("hbm_write_bytes", 0.0) if pk else None
    roofline = {"bound": "valu_issue" if (valu and valu["frac"] > achieved / HBM_PEAK_GBS) else "hbm",
                "kernel": ("k_shade_lean" if lean else "k_shade") if dom == "k_shade" else dom,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_hbm_low": achieved_low / HBM_PEAK_GBS, "frac_hbm_high": achieved / HBM_PEAK_GBS,
                "frac_valu": valu["frac"] if valu else None,
                "traffic": traffic_cal, "traffic_raw_counters": traffic_raw,
                "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE: every read request is 128 B on gfx950 (TCC_EA0_RDREQ_128B), FETCH_SIZE counts 64; L2 fabric side, Infinity-Cache hits included",
                "frac_fabric": (traffic_cal / (kernel_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic_cal and kernel_ms[dom] > 0) else None,
                "profile_tag": prof.get("tag") if prof else None,
                "algorithmic_bytes_per_launch": alg[dom], "algorithmic_bytes_per_launch_low": alg_low if dom == "k_shade" else alg[dom],
                "launch_ms": kernel_ms[dom], "launch_ms_from": f"hipEvents, mean of the last {len(profile_used)} of the {len(per_frame)} synchronised profile frames (the first ones run while the clocks ramp up)", "valu_issue": valu,
                "all_kernels_ms": kernel_ms, "all_kernels_algorithmic_bytes": alg}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_lib
        lut_rg = oracle_lib.brdf_lut(64, 64)
        mid = H // 2
        half = max(8, H // 8)
        cpu = cpu_baseline(scene, lut_rg, (max(0, mid - half), min(H, mid + half)), (mid - max(8, H // 16), mid + max(8, H // 16)))

    per_rank = None
    if world > 1:
        me = {"rank": rank, "device": local_rank, "rows": rows_mine, "loop_s": dt_local, "gather_alone_ms": gather_alone_ms,
              "kernel_ms": kernel_ms, "covered_pixels": st.get("covered_pixels"), "bin_entries": st.get("bin_entries"),
              "handoff_faults_in_loop": handoff_in_loop[0]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, me)
        dist.barrier()
    if world == 1:
        sharding_desc = "none"
    elif strips:
        sharding_desc = f"{world} row strips of {per} rows (+1-row halo for the MSAA edge detector) + RCCL {'gather to rank 0' if to_root else 'all-gather'} of the RGBA16F image, overlapped with the next frame"
    else:
        sharding_desc = (f"32-row bands round-robin over {world} ranks ({L} bands each) + RCCL {'gather to rank 0' if to_root else 'all-gather'} of the RGBA16F image + de-interleave; "
                         f"gather of frame i overlapped with the render of frame i+1 (double-buffered)"
                         + ("; MSAA: all-gather of the bands' boundary sample-0 keys between the geometry and the opaque pass" if args.msaa else ""))
    if rank == 0:
        out = {
            "metric": "frames/sec + shaded Mpix/s, 4K Sponza glTF, 1/2/4/8 MI355X",
            "value": fps, "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "frames_before_timed_loop": 1 + max(1, args.profile_frames),      # the first frame (uploads everything) + the synchronised per-kernel profile frames; then `warmup`, then `steps`
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (procedural scene generated in-repo; no glTF asset is available offline)",
            "shaded_mpix_per_s": W * H * fps / 1e6,
            "config": {"workload": f"{workload_name}: {n_tris} triangles, {len(scene.materials)} materials, "
                                   f"{len(scene.textures)} textures, {W}x{H}, geometry pass + opaque pass, " + ("MSAA x4 + edge resolve" if args.msaa else "single-sample") + (", MipmapMode::Gradient" + (" + max_anisotropy 16" if args.anisotropic else "") if args.mipmap else ", MipmapMode::None"),
                       "triangles": n_tris, "width": W, "height": H,
                       "sharding": sharding_desc,
                       "draws": len(r.host.draw_list()),
                       "frame_overlap": not args.no_overlap,
                       "stream_handoff": ("device flags (k_handoff_signal / k_handoff_wait)" if dev.stream_handoff() == 1 else "hipEvents" + (" (a device-flag gate timed out: measurement repeated)" if handoff_fault else "")) if not args.no_overlap else "none (one stream)",
                       "camera": "static" if args.static_camera else "orbit, one turn per 240 frames",
                       "environment": "uniform cubes (builder default)" if args.env == "uniform" else "texel cubemaps (procedural HDR, 256^2 x 9 levels + 32^2 irradiance)",
                       "opaque_route": "lean (k_shade_lean + k_shade_todo)" if lean else "general (k_shade)",
                       "library": os.path.relpath(hip_backend_path(), ROOT)},
            "frame_stats": {k: st[k] for k in ("triangles_in", "triangles_binned", "bin_entries", "covered_pixels", "shade_general_wavefronts",
                                                "frames_with_dropped_bin_entries", "bin_overflow_retries", "geometry_cache_blocks", "geometry_blocks")},
            "roofline": roofline,
            "cpu_baseline": cpu,
            # what crosses the boundary as host buffers: `value` is measured with the scene resident; a frame's own uploads are inside it
            "host_to_device": {"first_frame_bytes": first_frame_upload, "first_frame_ms": round(first_frame_ms, 2),
                               "texture_bytes_at_setup": int(sum(int(t.nbytes) for t in scene.textures)),
                               "steady_state_bytes_per_frame": steady_upload[0]},
            **({"collective": {"backend": "rccl" if backend == "nccl" else backend + " (rehearsal: every rank on GPU 0, staged through the host)",
                               "op": "gather to rank 0" if to_root else "all_gather_into_tensor", "bytes_per_rank": rows_out * W * 8,
                               "alone_ms": max(p["gather_alone_ms"] for p in per_rank), **collective_model(world, rows_out * W * 8, to_root), "launcher": "bench.py (child processes)" if os.environ.get("AWSM_BENCH_SELF_LAUNCHED") else "external (WORLD_SIZE was set)"},
                "per_rank": per_rank} if per_rank else {}),
            **({"check": check} if check else {}),
            **({"frame_trace": frame_trace} if frame_trace else {}),
        }
        print(json.dumps(out))
    dev.close()
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
