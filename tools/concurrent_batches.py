"""How much does a second batch in flight buy?  N host threads, each with its own UNet handle (same weights), its own HIP stream and its
own Generator / controller, run the SAME leg of bench.py (reverse or inversion + edit) on independent batches at the same time.

    python tools/concurrent_batches.py --arch sd15 --leg edit --batch 8 --threads 1,2,3 [--steps 6]

At the reference's shipped batch of 8 (running/sd1.5/launch_generation_iCD_sd1.5.sh:18) an SD1.5 evaluation is ~310 launches of 20 - 30 us
that fill half of the chip; independent batches (the next group of images) overlap their ramps and tails on the idle CUs."""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="sd15", choices=["sd15", "sdxl"])
    ap.add_argument("--leg", default="edit", choices=["reverse", "edit"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--threads", default="1,2")
    ap.add_argument("--steps", type=int, default=6, help="passes per thread count (total, split over the threads)")
    a = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    counts = [int(x) for x in a.threads.split(",")]
    W = bench.SD15Workload if a.arch == "sd15" else bench.SDXLWorkload
    wls = [W(dev) for _ in range(max(counts))]
    steps = [(w.edit_step(a.batch) if a.leg == "edit" else w.reverse_step(a.batch)) for w in wls]
    streams = [torch.cuda.Stream() for _ in wls]
    for st, s in zip(steps, streams):                     # warm-up on the stream the thread will use
        with torch.cuda.stream(s):
            st(); st()
    torch.cuda.synchronize()
    base = None
    for n in counts:
        per = max(1, a.steps // n)
        go = threading.Barrier(n + 1)

        def work(i):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                go.wait()
                for _ in range(per):
                    steps[i]()
        for rep in range(2):                              # second repetition is the one reported
            th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
            for t in th:
                t.start()
            torch.cuda.synchronize()
            go.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        rate = a.batch * per * n / dt
        base = base or rate
        print(f"{a.arch} {a.leg} B={a.batch}: {n} batch(es) in flight: {rate:8.2f} images/s  ({dt / per * 1e3:8.2f} ms per round of {n} x {a.batch} images, "
              f"{(rate / base - 1) * 100:+.1f} % vs one)", flush=True)


if __name__ == "__main__":
    main()
