#!/usr/bin/env python3
"""Per-kernel means of hardware counters from rocprofv3 --pmc passes (one or several counters per pass, csv output).

    rocprofv3 --pmc SQ_WAIT_ANY --kernel-trace --output-format csv -d out/p1 -o p -- python tools/gemm_bench.py attn 32 8 4096 4096 40
    python tools/pmc_kernel.py out [substring of the kernel name]

Prints, per kernel name containing the substring, the mean of every counter over its dispatches and the mean duration; with
SQ_WAVE_CYCLES present also each counter as a fraction of it (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES)."""
import collections, csv, glob, sys
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        k = k[:110]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, cs in acc.items():
    d = sorted(dur[k])
    print(f"{k}\n   dispatches {len(d)}  median duration {d[len(d) // 2]:.1f} us")
    wc = cs.get("SQ_WAVE_CYCLES")
    wcm = sum(wc) / len(wc) if wc else None
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print(f"   {c:28s} {m:16.0f}" + (f"   {m / wcm:7.3f} of SQ_WAVE_CYCLES" if wcm and c.startswith("SQ_") else ""))
