"""Process-group helpers with the reference's names (lib/utils/distributed.py:9-79). One process per GPU; the
'nccl' backend of PyTorch-ROCm IS RCCL, so collectives run over xGMI. Unlike the reference there is no
self-respawn under torch.distributed.launch: ranks are started by torchrun / `python -m torch.distributed.run`
and read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_distributed() else 1


def get_rank():
    return dist.get_rank() if is_distributed() else 0


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", 0))


def setup_process_group(backend=None):
    """env:// rendezvous (reference :71-79). backend defaults to nccl (=RCCL) with a GPU, gloo without."""
    if is_distributed() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(get_local_rank())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, init_method="env://")


def all_reduce_numpy(array):
    """reference :22-25 (metrics only)."""
    dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(array).to(dev)
    dist.all_reduce(t)
    return t.cpu().numpy()
