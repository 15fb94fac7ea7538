"""Several independent batches in flight on one GPU.

The reference's drivers process their prompt batches one after the other (running/sd1.5/generate.py:300-360, running/sd1.5/edit.py:
405-455, running/sdxl/generate.py:180-203).  On an MI355X one batch of the shipped sizes leaves CUs idle at every launch's ramp and tail
(8 images: ~310 launches of 20 - 30 us on half of the chip; 32 images / SDXL 8 images: the M = 8192 layers fill 62 - 84 % of it).  The NEXT
batch is independent work: `InFlight` runs N step callables - each bound to its own executor replica (`UNet2DConditionModel.replica()`:
same weights, own native handle / arena / caches / controller slot) - on N host threads and N HIP streams, all pulling passes from one
counter.  Every UNet call keeps its batch, results equal the sequential loop bit for bit; measured +6...8 % at the benchmark batch sizes
and +24 % / +31 % with two / three batches of 8 in flight (profiles/r04_in_flight.txt).  bench.py times its legs through this class.

    nets = [unet] + [unet.replica() for _ in range(n - 1)]
    flight = InFlight([make_step(net) for net in nets], device)      # make_step(net) -> callable running ONE batch on that executor
    outputs = flight.run(num_batches)                                 # in submission order; the caller's stream continues behind them
"""
import torch


class InFlight:
    """N independent batches in flight: one host thread + one HIP stream per executor replica, all pulling passes from one counter.
    An iCD evaluation at the reference's batch sizes leaves CUs idle at every launch's ramp and tail (8 images: ~310 launches of 20 - 30 us
    on half of the chip); the NEXT batch - independent samples, its own handle, arena and controller - fills them.  Results are those
    of the sequential loop (same kernels, same arithmetic per batch); only the wall clock changes."""

    def __init__(self, steps, device):
        self.steps, self.device = list(steps), device
        self.streams = [torch.cuda.Stream(device=device) for _ in self.steps] if len(self.steps) > 1 and torch.cuda.is_available() else None

    def __len__(self):
        return len(self.steps)

    def run(self, n):
        """n passes; returns their outputs in submission order (the caller's stream may consume them afterwards)."""
        if len(self.steps) == 1:
            return [self.steps[0]() for _ in range(n)]
        import contextlib
        import itertools
        import threading
        outs, errs, ticket = [None] * n, [], itertools.count()
        gpu = self.streams is not None                # (a CPU run - the gloo test - threads the same way, without streams)
        main = torch.cuda.current_stream() if gpu else None
        if gpu:
            for st in self.streams:
                st.wait_stream(main)                  # inputs produced on the caller's stream are visible to the side streams

        def work(i):
            try:
                if gpu:
                    torch.cuda.set_device(self.device)
                with (torch.cuda.stream(self.streams[i]) if gpu else contextlib.nullcontext()):
                    while True:
                        k = next(ticket)
                        if k >= n:
                            break
                        outs[k] = self.steps[i]()
            except BaseException as e:                # noqa: BLE001  (re-raised on the caller's thread)
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(self.steps))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        if gpu:
            for st in self.streams:
                main.wait_stream(st)                  # the caller's stream continues behind every side stream
        return outs
