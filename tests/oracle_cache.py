"""Hash-keyed fixtures of ORACLE outputs for the full-width parity cases (test infrastructure, like oracle/ itself).

The three full-width cases of the GPU suite spend 280 of its 670 - 860 s on the fp32 CPU oracle (12 - 13.5 TFLOP each, on whatever
cores the GPU box's host has free - the suite's run time moves by 40 % from box to box with it).  The oracle's output is a pure function
of (oracle sources, weights, inputs, schedule), all of them seeded CPU data: `lookup` hashes exactly those (xxh3-128 over every tensor's
dtype / shape / bytes, the scalars' repr and the text of oracle/unet_ref.py + oracle/sched_ref.py), and a file
tests/golden/oracle_cache/<tag>-<hash>.pt holding the oracle's result for that key replaces the recomputation.  Anything that changes the
key - another seed, another torch build whose CPU generator differs, an edit to the oracle - misses and the oracle runs live, as before:
a fixture can go stale only by missing, never by lying.

Made by (on a GPU box, because two of the keys contain a start latent the device computes):
    ICD_ORACLE_CACHE_WRITE=gpurun_out/oracle_cache python -m pytest tests/test_sdxl_full_gpu.py tests/test_sampler_gpu.py -m gpu -q \
        -k "full_width or full_sdxl_forward"
then copy gpurun_out/oracle_cache/*.pt to tests/golden/oracle_cache/.  ICD_ORACLE_CACHE_OFF=1 forces the live oracle everywhere.
Only the HIP path's REFERENCE side is cached; the product always runs."""
import os

import torch
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "golden", "oracle_cache")
_memo = {}


def _sources():
    if "src" not in _memo:
        h = xxhash.xxh3_128()
        for f in ("unet_ref.py", "sched_ref.py"):
            h.update(open(os.path.join(ROOT, "oracle", f), "rb").read())
        _memo["src"] = h.hexdigest()
    return _memo["src"]


def _feed(h, obj):
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu().contiguous()
        h.update(f"T{t.dtype}{tuple(t.shape)}".encode())
        if t.numel():
            h.update(t.view(torch.uint8).numpy() if t.dtype != torch.bool else t.to(torch.uint8).numpy())
    elif isinstance(obj, dict):
        h.update(b"D")
        for k in sorted(obj):
            h.update(str(k).encode())
            _feed(h, obj[k])
    elif isinstance(obj, (list, tuple)):
        h.update(b"L%d" % len(obj))
        for v in obj:
            _feed(h, v)
    elif hasattr(obj, "tobytes"):                         # numpy
        h.update(f"N{obj.dtype}{obj.shape}".encode())
        h.update(obj.tobytes())
    else:
        h.update(repr(obj).encode())


def weights_digest(sd):
    """Digest of a state dict, remembered per dict object (the full-width SDXL one is 10 GB of fp32: ~2 s)."""
    key = ("sd", id(sd), len(sd))
    if key not in _memo:
        h = xxhash.xxh3_128()
        _feed(h, sd)
        _memo[key] = h.hexdigest()
    return _memo[key]


def lookup(tag, sd, parts, compute):
    """The oracle's result for (tag, oracle sources, weights `sd`, `parts`): from the fixture if one with this exact key is committed,
    else `compute()` (and, with ICD_ORACLE_CACHE_WRITE=<dir>, saved there under the key's name)."""
    h = xxhash.xxh3_128()
    h.update(tag.encode())
    h.update(_sources().encode())
    h.update(weights_digest(sd).encode())
    _feed(h, parts)
    name = f"{tag}-{h.hexdigest()}.pt"
    path = os.path.join(DIR, name)
    if os.path.exists(path) and not os.environ.get("ICD_ORACLE_CACHE_OFF"):
        print(f"[oracle cache] {name}: committed fixture (key matches: same oracle sources, weights, inputs)")
        return torch.load(path, map_location="cpu")
    val = compute()
    out = os.environ.get("ICD_ORACLE_CACHE_WRITE")
    if out:
        os.makedirs(out, exist_ok=True)
        torch.save(val.detach().cpu().clone(), os.path.join(out, name))
        print(f"[oracle cache] wrote {os.path.join(out, name)}")
    return val
