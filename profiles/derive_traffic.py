#!/usr/bin/env python3
"""HBM bytes per launch of every libmmgnn kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
`--pmc X --kernel-trace`, same bench command), as MI355X_MICROARCH.md section HBM prescribes:

    bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024      (gfx950: FETCH_SIZE reports half of wide coalesced reads)

Dispatches are matched between the two passes by (kernel name, grid size): the launch sequence is deterministic.  Per
kernel the LARGEST launch shape (by grid size x bytes) is reported -- the patient-axis launches that bench.py's
`roofline` object describes -- together with the launch-weighted mean over all shapes.

usage: derive_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict

OP_OF_KERNEL = [("k_linear_bnbwd", "linear_bnbwd"), ("k_linear_fwd", "linear_fwd"), ("k_linear_small", "linear_fwd"), ("k_linear_wgrad", "linear_wgrad"),
                ("k_gather", "gather_rows"), ("k_scatter_strip", "scatter_rows"), ("k_scatter_units", "scatter_rows_rowscale"),
                ("k_scatter_bits", "scatter_rows"), ("k_scatter_mfma", "scatter_rows"),
                ("mmg_k_reduce_slabs<EpiScatter>", "scatter_reduce"), ("mmg_k_reduce_slabs<EpiStore>", "linear_wgrad_reduce"),
                ("k_pair_fwd", "pair_head_fwd"), ("k_pair_bwd", "pair_head_bwd"), ("k_bn_bwd_apply", "bn_bwd_apply"),
                ("k_col_reduce<1", "bn_bwd_stats"), ("k_col_reduce<0", "col_reduce2"), ("k_affine_act_drop", "affine_act_drop"),
                ("k_l2norm_fwd", "l2norm_fwd"), ("k_l2norm_bwd", "l2norm_bwd"), ("k_pair_loss", "pair_loss"),
                ("k_sel_", "pair_select")]


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


def load(path, counter):
    out = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        if "anonymous namespace" not in n and "mmg_k" not in n:
            continue
        out[(short(n), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def main(fetch_csv, write_csv, out_json):
    fetch, write = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    per_kernel = {}
    shapes = defaultdict(list)
    for key in fetch:
        if key not in write:
            continue
        f, w = statistics.median(fetch[key]), statistics.median(write[key])
        shapes[key[0]].append(dict(grid=key[1], launches=len(fetch[key]), fetch_kb=f, write_kb=w,
                                   hbm_bytes=int(2 * f * 1024 + w * 1024)))
    for k, lst in shapes.items():
        big = max(lst, key=lambda d: d["hbm_bytes"])
        tot_l = sum(d["launches"] for d in lst)
        per_kernel[k] = dict(hbm_bytes_per_launch=big["hbm_bytes"], grid=big["grid"], launches_profiled=big["launches"],
                             fetch_kb=big["fetch_kb"], write_kb=big["write_kb"],
                             mean_over_all_shapes=int(sum(d["hbm_bytes"] * d["launches"] for d in lst) / tot_l),
                             launches_all_shapes=tot_l)
    per_op = {}
    for k, v in per_kernel.items():
        for frag, op in OP_OF_KERNEL:
            if k.startswith(frag) or ("<" in frag and k.startswith(frag)):
                # an op is served by several kernel variants (prologue / accumulate / small-table instances): report the
                # one that moves the most bytes per step, which is the launch shape bench.py's roofline describes
                cur = per_op.get(op)
                if cur is None or v["hbm_bytes_per_launch"] * v["launches_profiled"] > \
                        cur["hbm_bytes_per_launch"] * cur["launches_profiled"]:
                    per_op[op] = dict(kernel=k, **v)
                break
    json.dump(dict(correction="HBM bytes = 2*FETCH_SIZE + WRITE_SIZE, counters in KB, separate --pmc passes "
                              "(MI355X_MICROARCH.md section HBM)",
                   note="per kernel: the largest launch shape (patient-axis launches); per_kernel is keyed by the "
                        "op name bench.py reports in roofline.kernel",
                   per_kernel=per_op, all_kernels=per_kernel), open(out_json, "w"), indent=1)
    for op, v in sorted(per_op.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{op:16s} {v['kernel'][:44]:44s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  ({v['launches_profiled']} launches)")


if __name__ == "__main__":
    main(*sys.argv[1:4])
