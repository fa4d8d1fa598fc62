"""Probe (round 4): the patient-sharded training step as ONE hipGraph with its RCCL all-reduces recorded inside, on one GPU
(world_size-1 `nccl` group) -- what `bench.py` times as `chain`.  Run under rocprofv3 --kernel-trace; the kernels between
the last two k_adam launches are one replay (profiles/r4_step_sequence_sharded_x100_d128.txt).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_sharded -- python3 profiles/probes/sharded_step.py
"""
import os
import socket
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.argv = [sys.argv[0], "--steps", "12", "--warmup", "2"] + sys.argv[1:]
import torch  # noqa: E402
import bench  # noqa: E402
from mmgnn import dist as mdist  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.distributed.init_process_group("nccl", store=torch.distributed.TCPStore("127.0.0.1", port, 1, True), rank=0,
                                     world_size=1, device_id=dev)
try:
    rec = bench.measure(args, 1, 0, dev, args.scale, False, args.dim, args.steps, args.warmup, False, force_comm=True)
    print("sharded step, world_size 1:", round(1e3 * rec["dt"] / args.steps, 3), "ms per step;", rec["launch"], file=sys.stderr)
finally:
    mdist.ShardComm().close()
