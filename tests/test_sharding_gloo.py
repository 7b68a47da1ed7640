"""N > 1 path on CPU: two processes (gloo), each renders its strip of the frame with the CPU oracle, the strips are
all-gathered with awsm_renderer_amd.sharding exactly as bench.py does, and the assembled image must equal the
single-process full frame bit for bit (a shard's rows are defined to be identical to the full frame's rows)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["AWSM_ROOT"])
import numpy as np, torch, torch.distributed as dist
from awsm_renderer_amd import scenes
from awsm_renderer_amd.sharding import strip_rows, gather_image
from oracle import oracle_lib
from tests import helpers
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
scene = scenes.atrium_scene(160, 101, detail=0.125, tex_scale=1 / 64)
lut = oracle_lib.brdf_lut(16, 16, threads=2)
model = helpers.build_model(scene)
y0, y1, per = strip_rows(scene.height, world, rank)
fr = helpers.oracle_frame(model, lut, rows=(y0, y1), threads=2)
# gloo has no 16-bit all-gather: ship each RGBA16F pixel as two int32 words (RCCL moves the f16 tensor directly)
strip = torch.zeros((per, scene.width, 2), dtype=torch.int32)
strip[: y1 - y0] = torch.from_numpy(np.ascontiguousarray(fr.rgba16f[y0:y1]).view(np.int32))
full = torch.zeros((world * per, scene.width, 2), dtype=torch.int32)
gather_image(strip, full, world)
keys = torch.zeros((per, scene.width), dtype=torch.int64)
keys[: y1 - y0] = torch.from_numpy(fr.keys[y0:y1].view(np.int64))
full_keys = torch.zeros((world * per, scene.width), dtype=torch.int64)
dist.all_gather_into_tensor(full_keys, keys)
if rank == 0:
    np.save(os.environ["AWSM_OUT"] + "_img.npy", full.numpy()[: scene.height])
    np.save(os.environ["AWSM_OUT"] + "_keys.npy", full_keys.numpy()[: scene.height])
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_strips_gather_to_the_full_frame(world, tmp_path):
    from awsm_renderer_amd import scenes
    from oracle import oracle_lib
    from tests import helpers
    out = str(tmp_path / "gather")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, AWSM_ROOT=ROOT, AWSM_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    scene = scenes.atrium_scene(160, 101, detail=0.125, tex_scale=1 / 64)
    lut = oracle_lib.brdf_lut(16, 16, threads=2)
    ref = helpers.oracle_frame(helpers.build_model(scene), lut, threads=4)
    img = np.ascontiguousarray(np.load(out + "_img.npy")).view(np.uint16)
    keys = np.load(out + "_keys.npy").view(np.uint64)
    assert np.array_equal(keys, ref.keys)
    assert np.array_equal(img, ref.rgba16f)


def test_strip_rows_cover_the_frame_exactly():
    from awsm_renderer_amd.sharding import strip_rows
    for h in (1, 31, 32, 33, 101, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            rows = [strip_rows(h, n, r) for r in range(n)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            for a, b in zip(rows, rows[1:]):
                assert a[1] == b[0] or (a[1] == h and b[0] == h)
            assert all(y1 - y0 <= per for y0, y1, per in rows)


BAND_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["AWSM_ROOT"])
import numpy as np, torch, torch.distributed as dist
from awsm_renderer_amd import scenes
from awsm_renderer_amd.sharding import band_rows, bands_per_rank, bands_to_image
from oracle import oracle_lib
from tests import helpers
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
scene = scenes.atrium_scene(160, 101, detail=0.125, tex_scale=1 / 64)
lut = oracle_lib.brdf_lut(16, 16, threads=2)
model = helpers.build_model(scene)
H, W = scene.height, scene.width
# a shard's rows are defined to equal the full frame's rows, so the CPU stand-in for "render my bands" is: full frame, keep mine
fr = helpers.oracle_frame(model, lut, threads=2)
rows = band_rows(H, world, rank)
L = bands_per_rank(H, world)
compact = torch.zeros((L * 32, W, 2), dtype=torch.int32)          # what awsm_hip_set_shard_bands(..., compact_output=1) writes
compact[: len(rows)] = torch.from_numpy(np.ascontiguousarray(fr.rgba16f[rows]).view(np.int32))
gathered = torch.zeros((world * L * 32, W, 2), dtype=torch.int32)
dist.all_gather_into_tensor(gathered, compact)
img = bands_to_image(gathered.view(world, L, 32, W, 2), H, world)
if rank == world - 1:
    np.save(os.environ["AWSM_OUT"] + "_img.npy", img.contiguous().numpy())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_bands_gather_to_the_full_frame(world, tmp_path):
    from awsm_renderer_amd import scenes
    from oracle import oracle_lib
    from tests import helpers
    out = str(tmp_path / "bands")
    script = tmp_path / "band_worker.py"
    script.write_text(BAND_WORKER)
    env = dict(os.environ, AWSM_ROOT=ROOT, AWSM_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    scene = scenes.atrium_scene(160, 101, detail=0.125, tex_scale=1 / 64)
    ref = helpers.oracle_frame(helpers.build_model(scene), oracle_lib.brdf_lut(16, 16, threads=2), threads=4)
    img = np.ascontiguousarray(np.load(out + "_img.npy")).view(np.uint16)
    assert np.array_equal(img, ref.rgba16f)


def test_band_rows_partition_the_frame():
    from awsm_renderer_amd.sharding import band_rows, bands_per_rank, bands_to_image
    for h in (1, 31, 32, 33, 101, 563, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            owned = [band_rows(h, n, r) for r in range(n)]
            assert sorted(y for rows in owned for y in rows) == list(range(h))
            L = bands_per_rank(h, n)
            assert all(len(rows) <= L * 32 for rows in owned)
            # layout check: put each row's index where the compact output would hold it and reassemble
            g = np.full((n, L * 32, 1, 1), -1, dtype=np.int64)
            for r, rows in enumerate(owned):
                g[r, : len(rows), 0, 0] = rows
            img = bands_to_image(g.reshape(n, L, 32, 1, 1), h, n)
            assert img[:, 0, 0].tolist() == list(range(h))


def test_bench_starts_its_own_ranks_and_never_touches_the_gpu_itself(tmp_path):
    """`python bench.py --gpus 2` as the driver types it (no launcher, WORLD_SIZE unset): the process must become a launcher — two children with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set — without importing torch (a process that has initialised the GPU must not start or become
    another program on this pool).  Without a GPU each rank refuses to run (no CPU fallback) and the launcher hands the failure on."""
    import torch
    if torch.cuda.device_count() > 0:      # (counting devices does not initialise the GPU; asked BEFORE anything is started: ADVICE r4)
        pytest.skip("a GPU is visible: the N > 1 path itself is what test_gpu_parity.py's rehearsal runs")
    probe = tmp_path / "sitecustomize.py"
    probe.write_text(
        "import os, sys, atexit\n"
        "def _report():\n"
        "    with open(os.environ['AWSM_PROBE_OUT'] + '.' + str(os.getpid()), 'w') as f:\n"
        "        f.write(repr({'rank': os.environ.get('RANK'), 'local_rank': os.environ.get('LOCAL_RANK'), 'world': os.environ.get('WORLD_SIZE'),\n"
        "                      'addr': os.environ.get('MASTER_ADDR'), 'port': os.environ.get('MASTER_PORT'), 'torch': 'torch' in sys.modules, 'argv': sys.argv}))\n"
        "atexit.register(_report)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PYTHONPATH=str(tmp_path) + os.pathsep + env.get("PYTHONPATH", ""), AWSM_PROBE_OUT=str(tmp_path / "probe"), AWSM_BENCH_RANK_GRACE_S="30")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stderr.count("needs an MI355X") == 2, p.stderr[-3000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    reports = [eval(f.read_text()) for f in tmp_path.glob("probe.*")]
    parent = [r for r in reports if r["rank"] is None]
    ranks = sorted((r for r in reports if r["rank"] is not None), key=lambda r: r["rank"])
    assert len(parent) == 1 and parent[0]["torch"] is False
    assert [(r["rank"], r["local_rank"], r["world"], r["addr"]) for r in ranks] == [("0", "0", "2", "127.0.0.1"), ("1", "1", "2", "127.0.0.1")]
    assert ranks[0]["port"] == ranks[1]["port"] and all(r["torch"] for r in ranks) and all(r["argv"][1:] == parent[0]["argv"][1:] for r in ranks)
