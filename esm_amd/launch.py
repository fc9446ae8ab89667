"""One process per GPU without an external launcher.

``python bench.py --gpus N`` and ``python -m esm_amd.extract ... --gpus N`` started from a plain shell re-execute
themselves under ``torch.distributed.run`` (one rank per GPU, rendezvous on 127.0.0.1 at a free port); started by a
launcher already (RANK / WORLD_SIZE in the environment) they just join it.  ``init_ranks`` is the one place where
the process group is created and checked: backend "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU
protocol tests.
"""
import os
import socket
import subprocess
import sys


def under_launcher() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch(n_ranks: int, target, argv) -> int:
    """Run ``target`` (a script path, or ("-m", "pkg.module")) with ``argv`` as ``n_ranks`` ranks on this node and
    return the launcher's exit code.  stdout / stderr are inherited, so rank 0's JSON line reaches the caller."""
    target = list(target) if isinstance(target, (tuple, list)) else [target]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + target + list(argv)
    return subprocess.call(cmd, env=env)


def init_ranks(expect_world: int, backend: str):
    """Join the launcher's process group.  Returns (dist or None, rank, world, local_rank); verifies that the world
    is the one asked for, that the backend is the one asked for, and that a collective actually works (an
    all-reduce of the rank ids)."""
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != expect_world:
        raise SystemExit(f"world size {world} does not match --gpus {expect_world}")
    if not under_launcher():
        return None, 0, 1, 0
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert dist.get_world_size() == expect_world and dist.get_backend() == backend, (dist.get_world_size(), dist.get_backend())
    probe = torch.tensor([float(rank)], device=dev)
    dist.all_reduce(probe)
    assert probe.item() == world * (world - 1) / 2, "all_reduce over the ranks returned a wrong sum"
    return dist, rank, world, local_rank
