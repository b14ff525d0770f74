"""Process-group helpers with the reference's names (lib/utils/distributed.py:9-79). One process per GPU; the
'nccl' backend of PyTorch-ROCm IS RCCL, so collectives run over xGMI.

Launch: like the reference, `main_contrastive.py --distributed --gpu 0 1 2 3` respawns itself with one rank per listed
GPU (`handle_distributed`, reference :27-69) -- here through `python -m torch.distributed.run` on 127.0.0.1 with a free
port instead of the deprecated torch.distributed.launch on the fixed port 29961 -- and ranks started by torchrun
directly (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) are accepted as they are."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized()


def exercise_single_rank():
    """CSEG_DIST_SINGLE_RANK=1: take the multi-rank code paths (SyncBN exchange, cross-rank contrast set, all-gather) even in
    a process group of ONE rank. A 1-GPU box cannot host two RCCL ranks, but it can host one: with this switch every
    collective of this code runs through RCCL on the device (tools/rccl_single_rank_check.py). Off by default: with one
    rank the exchanges are identities and only cost launches."""
    return os.environ.get("CSEG_DIST_SINGLE_RANK") == "1" and is_distributed()


def get_world_size():
    return dist.get_world_size() if is_distributed() else 1


def get_rank():
    return dist.get_rank() if is_distributed() else 0


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", 0))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched_by_torchrun():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ and "LOCAL_RANK" in os.environ


def respawn_command(n_ranks, script_argv, module=None):
    """argv of the launcher that starts `n_ranks` copies of this program on this node (rendezvous on 127.0.0.1: the
    container hostname may not resolve)."""
    cmd = [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    if module is not None:
        return cmd + ["-m", module] + list(script_argv)
    return cmd + list(script_argv)


def handle_distributed(args, main_file=None, module=None, argv=None):
    """reference :27-69. Not distributed: restrict the process to the listed GPUs. Distributed and already a rank
    (started by torchrun, or `--local_rank` given by a launcher): join the process group. Distributed and not yet a rank:
    start one rank per entry of `--gpu` (or per already-visible device) and exit with the launcher's return code."""
    visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    if not args.distributed:
        if launched_by_torchrun() and int(os.environ["WORLD_SIZE"]) > 1:
            setup_process_group()              # torchrun without --distributed: still one rank per process
        elif visible is None and args.gpu is not None and not torch.cuda.is_initialized():
            os.environ["HIP_VISIBLE_DEVICES"] = ",".join(map(str, args.gpu))
        return
    if launched_by_torchrun() or args.local_rank >= 0:
        if args.local_rank >= 0:
            os.environ.setdefault("LOCAL_RANK", str(args.local_rank))
        setup_process_group()
        return
    env = os.environ.copy()
    if visible is None:
        env["HIP_VISIBLE_DEVICES"] = ",".join(map(str, args.gpu))
        world = len(args.gpu)
    else:
        world = len(visible.split(","))
    argv = list(sys.argv[1:] if argv is None else argv)
    cmd = respawn_command(world, ([main_file] if module is None else []) + argv, module=module)
    print("World size:", world, flush=True)
    proc = subprocess.Popen(cmd, env=env)
    proc.wait()
    if proc.returncode != 0:
        raise subprocess.CalledProcessError(returncode=proc.returncode, cmd=cmd)
    sys.exit(0)


def setup_process_group(backend=None):
    """env:// rendezvous (reference :71-79). backend defaults to nccl (=RCCL) with a GPU, gloo without;
    CSEG_DIST_BACKEND overrides (gloo lets several ranks share one GPU for dry runs -- RCCL refuses that)."""
    if is_distributed() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if backend is None:
        backend = os.environ.get("CSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(get_local_rank() % torch.cuda.device_count())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, init_method="env://")


def device_index():
    """Device of this rank: LOCAL_RANK, folded onto the visible devices (several ranks may share a GPU under gloo)."""
    return get_local_rank() % max(1, torch.cuda.device_count()) if torch.cuda.is_available() else 0


def all_gather_cat(t):
    """[world * n, ...] = every rank's `t` (same shape on all ranks) stacked in rank order. RCCL all-gather on the
    'nccl' backend; on gloo with device tensors (no device all-gather there) each rank fills its own slot of a zero
    buffer and the slots are summed by all-reduce -- same bytes on the wire for the tiny buffers this path moves."""
    world = get_world_size()
    if world == 1 and not exercise_single_rank():
        return t
    t = t.contiguous()
    if dist.get_backend() == "nccl" or not t.is_cuda:
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t) if dist.get_backend() == "nccl" else \
            dist.all_gather(list(out.chunk(world)), t)
        return out
    out = torch.zeros((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    out[get_rank()] = t
    dist.all_reduce(out)
    return out.reshape((world * t.shape[0],) + tuple(t.shape[1:]))


def all_reduce_numpy(array):
    """reference :22-25 (metrics only)."""
    dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(array).to(dev)
    dist.all_reduce(t)
    return t.cpu().numpy()
