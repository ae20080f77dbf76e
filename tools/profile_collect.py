"""Turns the outputs of tools/profile_round.sh into the committed summaries:

    python tools/profile_collect.py r02

  profiles/<tag>_bench_cfg3_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (per kernel: calls, total, average)
  profiles/<tag>_pmc_fetch_size_summary.txt      mean FETCH_SIZE per launch and kernel (KiB as reported)
  profiles/<tag>_pmc_write_size_summary.txt      mean WRITE_SIZE per launch and kernel
  profiles/traffic.json                          HBM bytes per launch of the two bandwidth kernels, corrected as
                                                 MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled on gfx950 for 16-B-per-
                                                 lane reads, WRITE_SIZE as is), calibration lines included"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def find(d, pat):
    f = glob.glob(os.path.join(root, "gpurun_out", d, "**", pat), recursive=True)
    if not f:
        sys.exit(f"missing {d}/{pat}")
    return f[0]


shutil.copy(find(f"{tag}_stats", "*kernel_stats.csv"), os.path.join(out, f"{tag}_bench_cfg3_kernel_stats.csv"))

# The stats above average over the whole run, the 1000 untimed steps that build the sketch's hot set included (their churn
# shows in k_fix and in the one-kernel form of the update).  The same trace, last 300 launches of every kernel only:
trace = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_stats", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    per = defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(os.path.join(out, f"{tag}_bench_cfg3_kernel_stats_steady.csv"), "w") as f:
        f.write("Name,Calls,LastCalls,AverageNs,MinNs,MaxNs\n")
        rows = []
        for k, v in per.items():
            v.sort()
            d = [x[1] for x in v[-300:]]
            rows.append((sum(d), k, len(v), len(d), sum(d) / len(d), min(d), max(d)))
        for _, k, n, m, avg, lo, hi in sorted(rows, reverse=True):
            f.write(f"\"{k}\",{n},{m},{avg:.0f},{lo},{hi}\n")


def pmc(d, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, v in per.items():
        tail = v[-100:] if len(v) > 200 else v[len(v) // 2:]         # steady state: the last 100 launches (behind the preheat)
        res[k] = (sum(tail) / len(tail), len(v))
    return res


def write_summary(res, counter, path):
    with open(path, "w") as f:
        f.write(f"{'kernel':50s} {counter:>12s}   calls\n")
        for k, (m, n) in sorted(res.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            f.write(f"{k[:46]:50s} {m:12.0f}   {n}\n")


fetch, write = pmc(f"{tag}_fetch", "FETCH_SIZE"), pmc(f"{tag}_write", "WRITE_SIZE")
write_summary(fetch, "FETCH_SIZE", os.path.join(out, f"{tag}_pmc_fetch_size_summary.txt"))
write_summary(write, "WRITE_SIZE", os.path.join(out, f"{tag}_pmc_write_size_summary.txt"))


def pick(res, prefix):
    """mean per launch of the kernel whose name starts with `prefix`; of several instantiations, the one launched most"""
    best = None
    for k, v in res.items():
        name = k.replace("void ", "").replace("cafe::", "")
        if name.startswith(prefix) and (best is None or v[1] > best[1]):
            best = v
    return None if best is None else best[0]


F, B, D = 26, 8192, 128
alg = {"k_forward": F * B * (56 + 8 * D), "k_bwd_segments": F * B * 12 * D, "k_grad_norm": F * B * (4 * D + 4)}
tj = {"workload": "cfg3/b8192",
      "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_round.sh {tag}) of `bench.py --steps 6 "
                "--warmup 2 --preheat 1000` (no forced form of the update), per-launch mean of the last 100 launches of each kernel; FETCH_SIZE is in KiB and doubled (gfx950 "
                "tallies 128-B requests at 64 B for 16-B-per-lane reads, MI355X_MICROARCH.md HBM section; calibration: k_grad_norm reads "
                "exactly 109.05 MB), WRITE_SIZE in KiB taken as is (calibration: k_forward writes 109.05 MB + 1.7 MB)"}
for k in ("k_forward", "k_bwd_segments", "k_grad_norm"):
    fk, wk = pick(fetch, k), pick(write, k)
    if fk is None or wk is None:
        continue
    parts = None
    if k == "k_bwd_segments" and pick(fetch, "k_bwd_apply") is not None:
        # the update in two kernels (+ the compaction of the row list between them): their traffic is the update's
        parts = {}
        for extra in ("k_bwd_compact", "k_bwd_apply"):
            fe, we = pick(fetch, extra), pick(write, extra)
            if fe is None or we is None:
                continue
            parts[extra] = {"hbm_read_bytes": int(fe * 1024 * 2), "hbm_write_bytes": int(we * 1024)}
            fk += fe; wk += we
        parts["k_bwd_segments"] = {"hbm_read_bytes": int(pick(fetch, k) * 1024 * 2), "hbm_write_bytes": int(pick(write, k) * 1024)}
    rd, wr = int(fk * 1024 * 2), int(wk * 1024)
    tj[k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "total": rd + wr, "algorithmic": alg[k],
             "fetch_size_kib_reported": round(fk), "write_size_kib_reported": round(wk)}
    if parts:
        tj[k]["kernels"] = parts
json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(tj, indent=1))


# LDS bank conflicts (tools/profile_lds.sh): a pass of its own, summarised when its output is there
lds = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_lds", "**", "*counter_collection.csv"), recursive=True)
if lds:
    act, conf = pmc(f"{tag}_lds", "SQ_LDS_IDX_ACTIVE"), pmc(f"{tag}_lds", "SQ_LDS_BANK_CONFLICT")
    with open(os.path.join(out, f"{tag}_pmc_lds_summary.txt"), "w") as f:
        f.write(f"rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (tools/profile_lds.sh {tag}) of `bench.py --steps 6 --warmup 2 --preheat 1000`;\n"
                "per-launch mean of the last 100 launches of each kernel; conflict share = extra cycles / all LDS-array cycles\n\n")
        f.write(f"{'kernel':50s} {'LDS_IDX_ACTIVE':>17s} {'BANK_CONFLICT':>14s} {'share':>7s}  calls\n")
        for k, (a, n) in sorted(act.items(), key=lambda kv: -kv[1][0]):
            c = conf.get(k, (0.0, 0))[0]
            if a > 0:
                f.write(f"{k[:46]:50s} {a:17.0f} {c:14.0f} {100.0 * c / a:6.1f}%  {n}\n")
