"""Process-group bootstrap + the data-parallel sharding / gather of the iCD drivers (mirror of utils/dist_utils.py and
of the DP pieces of running/sd1.5/generate.py:29-39,372-397).

One process per GPU (torchrun), backend "nccl" which IS RCCL on ROCm; the only collective on the path is ONE all-gather
of the finished samples at the end of a run (uint8 images or fp16 latents + int64 ids) - samples are independent, the
U-Net loop itself never communicates.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def init(backend=None, timeout_s=None):
    """utils/dist_utils.py:8-22: env:// rendezvous with single-process defaults; binds the rank to its GPU.

    Every default only fills what the launcher left unset: a single-process run takes a port that is free right now (the
    reference's fixed 29500 fails next to a stale listener), HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, what RCCL needs on this
    driver stack) is set only when the variable is absent.  The collectives time out after ICD_DIST_TIMEOUT_S seconds (default
    600) instead of hanging a node when a rank died."""
    import datetime
    if 'MASTER_ADDR' not in os.environ and 'MASTER_PORT' not in os.environ:
        # no launcher's rendezvous to join: this process hosts its own store.  A stray TORCHELASTIC_USE_AGENT_STORE (a process started
        # FROM a torchrun worker inherits it) would make env:// a client of a store nobody runs, and the call would sit out its timeout.
        os.environ.pop('TORCHELASTIC_USE_AGENT_STORE', None)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('LOCAL_RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    if 'MASTER_PORT' not in os.environ:
        if int(os.environ['WORLD_SIZE']) > 1:
            # every rank must agree on the port, so nothing can be picked here: the reference's fixed default (utils/dist_utils.py:11-12)
            # keeps launchers that export only RANK / WORLD_SIZE / MASTER_ADDR (srun, mpirun wrappers) working - said out loud, because
            # a stale listener on 29500 is the usual reason such a run hangs at the rendezvous
            import warnings
            warnings.warn('dist_utils.init: WORLD_SIZE > 1 without MASTER_PORT - meeting on the reference\'s default port 29500')
            os.environ['MASTER_PORT'] = '29500'
        else:
            os.environ['MASTER_PORT'] = str(_free_port())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if not dist.is_initialized():
        t = float(timeout_s if timeout_s is not None else os.environ.get('ICD_DIST_TIMEOUT_S', '600'))
        dist.init_process_group(backend=backend, init_method='env://', timeout=datetime.timedelta(seconds=t))


def prepare_val_prompts(all_text, bs=20, max_cnt=5000):
    """Round-robin batch sharding of running/sd1.5/generate.py:29-39 (== running/sdxl/generate.py:25-35):
    ((N-1)//(bs*W)+1)*W batches via array_split, rank r takes batches r::W.  Returns (batches, index batches, texts)."""
    all_text = all_text[:max_cnt]
    W, r = get_world_size(), get_rank()
    num_batches = ((len(all_text) - 1) // (bs * W) + 1) * W
    batches = np.array_split(np.array(all_text), num_batches)[r::W]
    index = np.array_split(np.arange(len(all_text)), num_batches)[r::W]
    return batches, index, all_text


def gather_samples(local, local_ids, always_collective=False):
    """ONE all-gather of this rank's finished samples + their global ids (running/sd1.5/generate.py:372-378), then
    reorder by id (:386-397).  `local`: [N_local, ...] tensor (uint8 images or fp16 latents), `local_ids`: int64 [N_local].
    Every rank must contribute the same N_local (the reference has the same equal-shape requirement).
    Returns (samples ordered by global id, sorted ids) on every rank.  A single process skips the collective unless
    `always_collective` is set and a process group exists (the reference always calls all_gather; a world of one then runs the real
    RCCL all-gather on its one GPU - how the collective path is exercised on a 1-GPU box)."""
    W = get_world_size()
    if W == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        order = torch.argsort(local_ids)
        return local[order], local_ids[order]
    bufs = [torch.empty_like(local) for _ in range(W)]
    ids = [torch.empty_like(local_ids) for _ in range(W)]
    dist.all_gather(bufs, local.contiguous())
    dist.all_gather(ids, local_ids.contiguous())
    allx, alli = torch.cat(bufs, 0), torch.cat(ids, 0)
    order = torch.argsort(alli)
    return allx[order], alli[order]
