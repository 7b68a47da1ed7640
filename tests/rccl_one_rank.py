"""Run by tests/test_gpu_parity.py::test_rccl_one_rank_carries_what_the_library_rendered in a process of its own (an MI355X box).
bench.py's N > 1 loop with the collectives on RCCL — backend "nccl" IS RCCL on ROCm — at world size 1, which one GPU allows: the process group with
device_id, the library rendering into torch tensors on its own shade streams (frames overlapped), awsm_hip_frame_flush, all_gather_into_tensor /
gather to root / all_reduce / barrier on device tensors, double-buffered.  What the gathers deliver must be the frame, bit for bit.  Prints "ok"."""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch
import torch.distributed as dist
from awsm_renderer_amd import scenes
from awsm_renderer_amd.host import Renderer
from awsm_renderer_amd.hip_backend import HipDevice

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
W, H = 640, 363
scene = scenes.atrium_scene(W, H, detail=0.125, tex_scale=1 / 16)
stream = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream)
r = Renderer(scene, device=0, stream=stream.cuda_stream, lut_size=64, overlap_frames=True)
dev = HipDevice.from_ctx(r.host.device_ctx, W, H)
r.render(sync=True)
ref = torch.from_numpy(dev.read_opaque().copy()).cuda().view(torch.int16)          # the library's own image of this (static) frame
mine = [torch.zeros((H, W, 4), dtype=torch.float16, device="cuda") for _ in range(2)]
gathered = [torch.zeros((1, H, W, 4), dtype=torch.float16, device="cuda") for _ in range(2)]
pending = [None, None]
bad = 0
for i in range(8):
    b = i % 2
    if pending[b] is not None:
        pending[b].wait()
        bad += int(not torch.equal(gathered[b][0].view(torch.int16), ref.view(H, W, 4)))
        gathered[b].zero_()
    mine[b].zero_()
    dev.bind_output(mine[b].data_ptr(), H * W * 8)
    r.host.render(sync=False)
    dev.frame_flush()                      # the opaque pass ran on the library's shade stream: ordered before what this stream enqueues next
    if i % 4 == 3:
        pending[b] = dist.gather(mine[b], gather_list=list(gathered[b].unbind(0)), dst=0, async_op=True)
    else:
        pending[b] = dist.all_gather_into_tensor(gathered[b].view(H, W, 4), mine[b], async_op=True)
for b in range(2):
    pending[b].wait()
    bad += int(not torch.equal(gathered[b][0].view(torch.int16), ref.view(H, W, 4)))
t = torch.ones(1, dtype=torch.int32, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
assert int(t.item()) == 1 and bad == 0, (int(t.item()), bad)
r.host.render(sync=True)
dist.destroy_process_group()
print("ok")
