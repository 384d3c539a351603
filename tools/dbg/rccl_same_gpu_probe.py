"""Can two ranks of an RCCL ("nccl") process group share ONE GPU on this box? If so, bench.py's device hand-off (a tensor view of a
libmp2gpu allocation sent point to point) can be exercised over RCCL without a second GPU. Run under torch.distributed.run with 2 ranks.

Answer on the round-6 box (RCCL 2.26.6): no -- init_process_group / the first collective fails with "NCCL error ... invalid usage"
(two ranks on one device are refused), so the device hand-off between two GPUs stays unexercised until the driver's SCALE run; what
bench.py does about that: sharding.probe_handoff before t0, with the staged fallback."""
import importlib, os, sys, datetime
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
    one = torch.ones(1, dtype=torch.int64, device="cuda")
    dist.all_reduce(one)
    print(f"rank {rank}: all_reduce over RCCL with both ranks on GPU 0 -> {int(one.item())}", flush=True)
    mp2 = importlib.import_module("mapreduce-plonky2_amd")
    sh = importlib.import_module("mapreduce-plonky2_amd.sharding")
    ctx = mp2.Context(0)
    dev = torch.device("cuda", 0)

    def direct(words):
        buf = ctx.to_device(words)
        ctx.sync()
        return torch.as_tensor(sh._RawView(buf.ptr.value, len(words), buf), device=dev)

    def staged(words):
        return torch.from_numpy(np.ascontiguousarray(words, dtype=np.uint64).view(np.int64).copy()).to(dev)
    print(f"rank {rank}: probe_handoff -> {sh.probe_handoff(dist, direct, staged, dev)}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)
