"""Turns the rocprofv3 outputs of collect_r4.sh <tag> into the small files kept under profiles/ (run on the GPU box;
the files land in gpurun_out/r4_<tag>/keep/ and are copied to profiles/ afterwards).
  r4_kernel_stats_<tag>{,_graph}.txt/.csv : rocprofv3 --kernel-trace --stats, eager / replayed hipGraph
  r4_step_sequence_<tag>.txt             : the kernels of ONE replayed step in start order (start, duration, gap, queue)
  r4_traffic_<tag>.json                   : HBM bytes per launch, 2 * FETCH_SIZE + WRITE_SIZE from separate PMC passes
  r4_pmc_<tag>.json                       : matrix-core / VALU / LDS counters per kernel (largest launch shape)"""
import csv, glob, json, os, re, statistics, subprocess, sys
from collections import defaultdict

TAG = sys.argv[1]
O = f"gpurun_out/r4_{TAG}"
KEEP = f"{O}/keep"
os.makedirs(KEEP, exist_ok=True)


def find(d, pat):
    f = glob.glob(f"{O}/{d}/**/*{pat}", recursive=True)
    return f[0] if f else None


def short(n):
    return re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))


def stats(d, out_txt, out_csv, top=90):
    f = find(d, "kernel_stats.csv")
    if not f:
        return
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out_txt, "w") as o:
        for r in rows[:top]:
            o.write(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}%  calls {int(r['Calls']):5d}  "
                    f"avg {float(r['AverageNs'])/1e3:8.1f} us  {short(r['Name'])[:110]}\n")
        o.write(f"total {tot/1e6:.3f} ms\n")
    with open(out_csv, "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([short(r["Name"])[:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])


def seq(d, out):
    f = find(d, "kernel_trace.csv")
    if not f:
        return
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
    if len(idx) < 3:
        return
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a + 1]["Start_Timestamp"])
    prev = t0
    with open(out, "w") as o:
        for r in rows[a + 1:b + 1]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o.write(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-prev)/1e3:6.1f}  q{r.get('Queue_Id','?')}  {short(r['Kernel_Name'])[:120]}\n")
            prev = max(prev, e)
        o.write(f"kernels {b-a}  span {(int(rows[b]['End_Timestamp'])-t0)/1e3:.1f} us\n")


def pmc(dirs, out):
    agg = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))      # kernel -> grid -> counter -> values
    for d in dirs:
        f = find(d, "counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]).split("(")[0]
            if not (k.startswith("k_") or k.startswith("mmg_k_")):
                continue
            agg[k][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        return
    res = {}
    for k, grids in agg.items():
        g = max(grids, key=lambda g: g * len(next(iter(grids[g].values()))))          # the shape that does the most work
        res[k] = {"grid": g, **{c: statistics.median(v) for c, v in sorted(grids[g].items())},
                  "launches_seen": len(next(iter(grids[g].values())))}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


stats("stats_eager", f"{KEEP}/r4_kernel_stats_{TAG}.txt", f"{KEEP}/r4_kernel_stats_{TAG}.csv")
stats("stats_graph", f"{KEEP}/r4_kernel_stats_{TAG}_graph.txt", f"{KEEP}/r4_kernel_stats_{TAG}_graph.csv")
seq("stats_graph", f"{KEEP}/r4_step_sequence_{TAG}.txt")
fa, fw = find("pmc_fetch", "counter_collection.csv"), find("pmc_write", "counter_collection.csv")
if fa and fw:
    subprocess.run([sys.executable, "profiles/derive_traffic.py", fa, fw, f"{KEEP}/r4_traffic_{TAG}.json"], check=False)
pmc(["pmc_a", "pmc_b", "pmc_c"], f"{KEEP}/r4_pmc_{TAG}.json")
print("done:", sorted(os.listdir(KEEP)))
