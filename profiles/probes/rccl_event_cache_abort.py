"""Probe (round 4): the abort of a sharded run inside ProcessGroupNCCL's watchdog thread
  "HIP error: operation not permitted on an event last recorded in a capturing stream" (hipErrorCapturedEvent)
and what makes it go away.  Run on the GPU box, one mode per process (an abort ends the process):
    python profiles/probes/rccl_event_cache_abort.py cache          # torch's default: ProcessGroupNCCL reuses its events
    python profiles/probes/rccl_event_cache_abort.py cache_other    # same, second capture on ANOTHER stream
    python profiles/probes/rccl_event_cache_abort.py nocache        # TORCH_NCCL_CUDA_EVENT_CACHE=0
    python profiles/probes/rccl_event_cache_abort.py eager_stream   # the eager all-reduces on mmgnn.dist's own stream (the fix)
Measured (gpurun_out/r4t, round 4): cache -> abort; nocache -> abort; cache_other -> survives: the event cache is not the
cause, the stream the eager collectives' events were recorded on must not be capturing while the watchdog still polls them.
Sequence: (1) an all-reduce RECORDED into a hipGraph on stream S: its end event is recorded inside the capture and goes back
to the process group's event cache when the call returns; (2) eager all-reduces: they take events from that cache and are
handed to the watchdog thread, which polls them (hipEventQuery) every 100 ms; (3) a second capture on S that lasts longer than
one poll.  If HIP refuses the query of an event that was ONCE recorded in a capture on a stream that is capturing NOW, the
watchdog throws and the process aborts in (3)."""
import os
import socket
import sys
import time

mode = sys.argv[1] if len(sys.argv) > 1 else "cache"
if mode == "nocache":
    os.environ["TORCH_NCCL_CUDA_EVENT_CACHE"] = "0"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", store=dist.TCPStore("127.0.0.1", port, 1, True), rank=0, world_size=1, device_id=dev)
t = torch.zeros(64, device=dev)
S = torch.cuda.Stream()
S2 = torch.cuda.Stream()
with torch.cuda.stream(S):
    dist.all_reduce(t)
    torch.cuda.synchronize()
    time.sleep(0.5)
    g = torch.cuda.CUDAGraph()
    g.capture_begin(capture_error_mode="thread_local")
    dist.all_reduce(t)                       # (1)
    g.capture_end()
    if mode == "eager_stream":
        sys.path.insert(0, os.getcwd())
        import mmgnn  # noqa: F401
        from mmgnn import dist as md
        comm = md.ShardComm()
        for _ in range(20):
            comm.raw_all_reduce(t)           # (2) on the library's stream, ordered against S by events
    else:
        for _ in range(20):
            dist.all_reduce(t)               # (2)
with torch.cuda.stream(S2 if mode == "cache_other" else S):
    g2 = torch.cuda.CUDAGraph()
    g2.capture_begin(capture_error_mode="thread_local")
    t.add_(1)
    time.sleep(0.5)                          # (3)
    g2.capture_end()
torch.cuda.synchronize()
g2.replay(); g.replay()
torch.cuda.synchronize()
del g, g2
dist.destroy_process_group()
print("survived:", mode, flush=True)
