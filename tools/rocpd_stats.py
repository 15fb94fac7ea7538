#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / % - the `--stats` table.

    python tools/rocpd_stats.py gpurun_out/prof/run_results.db > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    span = max(r[2] for r in rows) - min(r[1] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# {len(rows)} dispatches, sum of kernel durations {total / 1e6:.3f} ms, first-start..last-end span {span / 1e6:.3f} ms")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:110s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / total:6.2f}")
    # roll-up into the families bench.py's in-process profiler reports ("roofline.kernel" / "kernel_families")
    fam = {}
    for n, a in agg.items():
        m = re.search(r"gemm(?:_big|_pp320|_pp)?_kernel<(\d)", n)
        mx = re.search(r"gemm_kernel<\d, (?:false|true), \d, \d, true", n) or re.search(r"gemm_big_kernel<\d, \d, \d, \d, \d, true", n)
        if mx:
            f = "xattn_fused"
        elif m:
            f = "gemm_dense" if m.group(1) == "0" else "gemm_conv"
        elif "attn_probs" in n:
            f = "softmax (one-pass probabilities)"
        elif "splitk_reduce" in n:
            f = "splitk_reduce (bench.py counts it inside the GEMM family that launched it)"
        elif "attn_fused" in n or "attn_cross" in n:
            f = "attn_fused"
        elif "gn_" in n:
            f = "groupnorm"
        elif "layernorm" in n:
            f = "layernorm"
        else:
            f = "other (elementwise, torch)"
        x = fam.setdefault(f, [0, 0])
        x[0] += a[0]; x[1] += a[1]
    print()
    print(f"{'family':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for f, x in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{f:110s} {x[0]:7d} {x[1] / 1e6:10.3f} {x[1] / x[0] / 1e3:10.2f} {100 * x[1] / total:6.2f}")


    # idle time between consecutive kernels of the product library (start[i+1] - end[i], overlaps count as 0), restricted to the
    # steady region: pairs whose gap is < 50 us (longer gaps are host-side pauses between steps / phases of the script)
    ours = sorted((s_, e_, short(n)) for n, s_, e_ in rows if re.search(r"gemm|attn|gn_|layernorm|conv_out|x0_step|sinusoid|silu|geglu|softmax|upsample|embed", n))
    gaps, busy = [], 0
    for (s0, e0, _), (s1, e1, _) in zip(ours, ours[1:]):
        g = s1 - e0
        if g < 50_000:
            gaps.append(max(g, 0)); busy += e1 - s1
    if gaps:
        gaps.sort()
        tot = sum(gaps)
        print()
        print(f"# inter-kernel gaps (consecutive product kernels, gaps < 50 us): {len(gaps)} pairs, mean {tot / len(gaps) / 1e3:.2f} us, "
              f"median {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us; idle {tot / 1e6:.2f} ms beside "
              f"{busy / 1e6:.2f} ms of kernels = {100 * tot / (tot + busy):.1f} % of the stream")


if __name__ == "__main__":
    main(sys.argv[1])
