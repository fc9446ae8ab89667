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


def rank_cpu_slice(local_rank: int, local_world: int, cpus=None):
    """CPUs for one rank of ``local_world`` ranks on this host: a contiguous 1 / local_world share of the allowed
    hardware threads, hyper-thread siblings kept together (Linux numbers the second thread of core i as i + n_cores),
    so that a rank's tokeniser, writer threads and the driver thread of its GPU stay on neighbouring cores (one NUMA
    node when the shares align) instead of migrating over 256 threads."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    n = len(cpus)
    if local_world <= 1 or n < 2 * local_world:
        return set(cpus)
    sib = {}
    try:  # thread_siblings_list of the first allowed CPU tells the numbering scheme
        with open(f"/sys/devices/system/cpu/cpu{cpus[0]}/topology/thread_siblings_list") as fh:
            txt = fh.read().strip()
        first = sorted(int(x) for part in txt.split(",") for x in (range(int(part.split("-")[0]), int(part.split("-")[-1]) + 1)))
        # two hardware threads per core, numbered [a, a + h) and their siblings [b, b + h): the whole host (b = a + h) or one
        # NUMA node's set (e.g. 0-63 + 128-191)
        h = n // 2
        if (len(first) == 2 and first[0] == cpus[0] and cpus[:h] == list(range(cpus[0], cpus[0] + h))
                and cpus[h:] == list(range(first[1], first[1] + h))):
            sib = {c: c + (first[1] - first[0]) for c in cpus[:h]}
    except (OSError, ValueError):
        pass
    if sib:  # cores [a, b) and their siblings
        cores = cpus[: n // 2]
        per = len(cores) // local_world
        mine = cores[local_rank * per:(local_rank + 1) * per]
        return set(mine) | {sib[c] for c in mine}
    per = n // local_world
    return set(cpus[local_rank * per:(local_rank + 1) * per])


def pin_rank_cpus(local_rank: int, local_world: int, force: bool = False, peers=None):
    """Restrict this process (and the threads it starts) to its share of the host's CPUs.  ``ESM_AMD_NO_AFFINITY=1``
    switches it off.  Returns the CPU set in effect.

    ``peers``: the affinity sets of ALL local ranks (index = local rank; init_ranks gathers them once the process group is
    up).  With them the decision is exact: ranks whose sets are identical share that set and it is sliced among exactly
    those ranks (all ranks of a docker / Slurm cpuset, or the 4 ranks of one NUMA node under torchrun's NUMA binding); a
    set nobody else has is a per-rank set and is kept (the size of a set alone cannot tell a node's set from a
    job's).  Without ``peers`` (no process group yet / a single caller): a set of at most one rank's share of the host is
    kept, a larger restricted set is taken as shared by all local ranks, the whole host is sliced
    by local_world.  ``force`` slices by local_world regardless."""
    if os.environ.get("ESM_AMD_NO_AFFINITY", "0") == "1" or not hasattr(os, "sched_setaffinity"):
        return set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set()
    have = os.sched_getaffinity(0)
    host = os.cpu_count() or len(have)
    if peers is None and not force and local_world > 1 and len(have) <= host // local_world:
        return set(have)  # a per-rank set from whoever started us: not ours to narrow further
    if peers is not None and not force:
        same = [r for r in range(len(peers)) if set(peers[r]) == set(peers[local_rank])]
        if len(same) <= 1:
            return set(have)
        want = rank_cpu_slice(same.index(local_rank), len(same), cpus=have)
    else:
        want = rank_cpu_slice(local_rank, local_world, cpus=have)
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        pass
    return set(os.sched_getaffinity(0))


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

    # one node: every rank gets its own share of the host's CPUs (tokeniser + writer threads + GPU driver thread)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))

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
    # every rank's share of the host's CPUs, decided on the affinity sets of all local ranks (one node: world == local_world)
    peers = None
    if hasattr(os, "sched_getaffinity") and world == local_world:
        try:  # a plain tensor all-gather of CPU bit masks (the collective path the probe above just exercised), never fatal
            ncpu = 4096
            mine = torch.zeros(ncpu, dtype=torch.uint8)
            for c in os.sched_getaffinity(0):
                if c < ncpu:
                    mine[c] = 1
            mine = mine.to(dev)
            masks = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(masks, mine)
            peers = [torch.nonzero(m.cpu()).flatten().tolist() for m in masks]
        except Exception:
            peers = None
    cpus = pin_rank_cpus(local_rank, local_world, peers=peers)
    if cpus:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
    return dist, rank, world, local_rank
