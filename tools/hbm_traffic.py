#!/usr/bin/env python3
"""Per-launch HBM-side traffic of each kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/F -o p -- python bench.py ... --no-profile
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/W -o p -- python bench.py ... --no-profile
    python tools/hbm_traffic.py out/F out/W --arch sd15 --batch 32 > profiles/rNN_hbm_traffic_sd15_b32.json

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE counts 128-B requests at 64 B for wide
(16 B / lane) coalesced reads, global_load and global_load_lds alike -> doubled; WRITE_SIZE is used as reported
(calibrated here on a conv with a known 83.9 MB output: 1.08x).  Infinity-Cache hits are included in both, so this is
an upper bound on DRAM bytes.
"""
import argparse, collections, csv, glob, hashlib, json, os, re


def kernels_sha():
    """Same digest bench.py stamps into roofline.kernels_sha: which kernel sources this PMC summary was taken on."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "invertible_cd_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "*"))):
        if f.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def family(k):
    """Kernel name -> the family names of bench.py's in-process profiler (include/icd_amd.h ICD_PROF_*)."""
    m = re.search(r"gemm_kernel<(\d), (?:false|true), \d, \d, (false|true)", k)
    if m and m.group(2) == "true":
        return "xattn_fused"
    if re.search(r"gemm_big_kernel<\d, \d, \d, \d, \d, true", k):        # 256 x 256 host of the fused query-projection + attention
        return "xattn_fused"
    m = re.search(r"gemm(?:_big|_pp320|_pp)?_kernel<(\d)", k)
    if m:
        return "gemm_dense" if m.group(1) == "0" else "gemm_conv"
    for pat, f in (("splitk_reduce", "splitk_reduce"), ("attn_probs", "softmax"), ("attn_fused", "attn_fused"), ("attn_cross", "attn_fused"),
                   ("gn_", "groupnorm"), ("layernorm", "layernorm"), ("softmax", "softmax")):
        if pat in k:
            return f
    return None


def collect(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam:
                agg[fam][0] += 1
                agg[fam][1] += float(r["Counter_Value"])
    return agg


ap = argparse.ArgumentParser()
ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
ap.add_argument("--arch", required=True); ap.add_argument("--batch", type=int, required=True)
a = ap.parse_args()
F, W = collect(a.fetch_dir), collect(a.write_dir)
out = {"arch": a.arch, "per_gpu_batch": a.batch, "unit": "bytes per launch", "kernels_sha": kernels_sha(),
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "families": {}}
for fam in sorted(set(F) | set(W)):
    nf, vf = F.get(fam, [0, 0.0]); nw, vw = W.get(fam, [0, 0.0])
    fetch = 2.0 * vf / nf * 1024 if nf else 0.0
    write = vw / nw * 1024 if nw else 0.0
    out["families"][fam] = {"launches_sampled": nf or nw, "fetch_bytes": round(fetch), "write_bytes": round(write),
                            "traffic_bytes": round(fetch + write)}
print(json.dumps(out, indent=1))
