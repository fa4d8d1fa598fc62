"""Probe (round 4): hipMemsetAsync recorded into a hipGraph by stream capture, replayed.

For a list of sizes: capture {hipMemsetAsync(z, 0, 4 n) ; out.copy_(z)} on a side stream, then four times: fill z with 7.0
eagerly, replay, count the elements of `out` that are not 0.  Measured on MI355X with the HIP runtime PyTorch 2.10+rocm7.0
loads (gpurun_out/r4x/memset.log, round 4): every SECOND graph of the process replays its memset node with the 16-byte
pattern {n, 1, 0, 0} (int32) instead of zeros -- half of the buffer's dwords non-zero -- whatever n is (1 .. 2^20 + 1).
The library therefore zero-fills with a kernel (mmg_zero_async, csrc/common.h); this probe calls the runtime directly.

    python profiles/probes/hipgraph_memset_node.py
"""
import ctypes
import os
import sys

import torch


def hip_runtime():
    for line in open("/proc/self/maps"):
        path = line.split()[-1]
        if "libamdhip64" in os.path.basename(path):
            return ctypes.CDLL(path)
    raise RuntimeError("no HIP runtime mapped into this process")


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)
    hip = hip_runtime()
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    sizes = [5829, 5824, 1, 3, 63, 64, 65, 1000, 8064, 16512, 100000, 1 << 20, (1 << 20) + 1]
    n_bad = 0
    for i, n in enumerate(sizes):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = torch.zeros(n, device=dev)
            g = torch.cuda.CUDAGraph()
            g.capture_begin()
            z = torch.empty(n, device=dev)
            rc = hip.hipMemsetAsync(z.data_ptr(), 0, 4 * n, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            out.copy_(z)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        worst = None
        for rep in range(4):
            z.fill_(7.0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            nz = int((out != 0).sum())
            if nz and worst is None:
                worst = (rep, nz, out.view(torch.int32)[:8].tolist())
        print(f"graph {i:2d}  n = {n:8d}: " + ("zeros at every replay" if worst is None else
              f"replay {worst[0]}: {worst[1]} non-zero elements, first dwords {worst[2]}"), flush=True)
        n_bad += worst is not None
        del g
    print(f"{n_bad} of {len(sizes)} graphs replayed a memset node that did not zero its buffer")
    return 0


if __name__ == "__main__":
    sys.exit(main())
