"""Multi-GPU sharding: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

Self-play games never interact (the reference runs them in separate processes,
training_pipeline.py:325-329), so workers are sharded by contiguous id blocks
with NO per-step communication; the only collective is one gather of the
finished compact tuples / game results to rank 0 at the end (replacing the
reference's pickle-per-process + merge_data, training_pipeline.py:277-284).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def local_device(local_rank):
    """cuda device of this rank.  More ranks than GPUs (the 2-rank gloo test on a 1-GPU box) share devices."""
    return torch.device("cuda", int(local_rank) % max(1, torch.cuda.device_count()))


def init_from_env(backend=None):
    """Initialise the default process group when launched under torchrun.  Backend: the argument,
    else $CKR_DIST_BACKEND, else RCCL ("nccl") when a GPU is present, gloo otherwise.  With gloo the
    gather payloads are staged through host memory (gather_rows)."""
    rank, local_rank, world = env_world()
    # (CKR_FORCE_COLLECTIVE: a world of ONE rank still builds the process group, so that a 1-GPU box runs the job's collective
    # through RCCL -- bench.py under `torchrun --nproc-per-node 1`)
    if (world > 1 or (os.environ.get("CKR_FORCE_COLLECTIVE") and "RANK" in os.environ)) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("CKR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index):
    """NUMA node of a GPU from sysfs (PCI bus id -> /sys/bus/pci/devices/<id>/numa_node); None when the host does not say."""
    try:
        bus = torch.cuda.get_device_properties(int(device_index)).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(int(device_index)), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(int(device_index)), "pci_device_id", 0)
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (dom, bus, dev)
        with open(path) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def pin_to_gpu(local_rank, ranks_on_node=1, cpu_threads=1):
    """Host placement of one rank (the reference gives every worker process a core of its own, training_pipeline.py:325-329; here a
    rank is ONE Python thread replaying HIP graphs, and an unpinned thread wanders between the sockets of an 8-GPU node): restrict
    the process to cores of its GPU's NUMA node -- the node's cores are divided between the ranks whose GPUs share it -- or, when
    sysfs names no node, to an equal share of the cores the process may use.  Returns what was chosen, for the bench line:
    {"numa_node", "cpus" (count), "first_cpu", "last_cpu", "pinned"}.  CKR_NO_PIN=1: report only.
    cpu_threads: PyTorch's CPU thread count from here on (None: left alone).  A rank's host work is graph replays plus a few small
    tensor operations when a job is set up (building a network, packing its weights); with the process pinned, PyTorch's pool of
    one thread per core of the WHOLE host turns each of them into a pile-up -- a 400-game tournament's set-up 0.3 -> 3.9 s, measured
    with tools/small_jobs_context_check.py -- and one thread is the fastest setting even unpinned (6.4 against 6.7 s)."""
    info = {"numa_node": None, "cpus": None, "first_cpu": None, "last_cpu": None, "pinned": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return info
    dev = local_device(local_rank).index if torch.cuda.is_available() else None
    node = gpu_numa_node(dev) if dev is not None else None
    share = None
    if node is not None:
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                cpus = sorted(_parse_cpulist(f.read()) & set(allowed))
            # ranks whose GPUs sit on the same node split its cores
            n_gpu = max(1, torch.cuda.device_count())
            same = [r for r in range(int(ranks_on_node)) if gpu_numa_node(r % n_gpu) == node]
            k = same.index(int(local_rank)) if int(local_rank) in same else 0
            per = max(1, len(cpus) // max(1, len(same)))
            share = cpus[k * per:(k + 1) * per] or cpus
        except Exception:
            share = None
    if not share:
        per = max(1, len(allowed) // max(1, int(ranks_on_node)))
        k = int(local_rank) % max(1, int(ranks_on_node))
        share = allowed[k * per:(k + 1) * per] or allowed
    info.update(numa_node=node, cpus=len(share), first_cpu=share[0], last_cpu=share[-1])
    if os.environ.get("CKR_NO_PIN") != "1":
        try:
            os.sched_setaffinity(0, share)
            info["pinned"] = True
            if cpu_threads:
                torch.set_num_threads(int(cpu_threads))
                info["torch_threads"] = int(cpu_threads)
        except OSError:
            pass
    return info


def shard_range(n_workers, rank, world):
    """Contiguous block of worker ids owned by `rank`: (first, count)."""
    base, rem = divmod(int(n_workers), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_rows(local, dst=0, force_collective=False):
    """Gather variable-length row blocks (2-D tensors with equal row width) to
    `dst`: sizes via one all_gather, payload via one gather of padded blocks.
    Returns the concatenation on dst, None elsewhere.  Works on the device the
    backend expects (cuda for RCCL, cpu for gloo).  A single-rank job issues no
    collective unless force_collective is set (tests: the RCCL branch on a 1-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    home = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:        # gloo moves host memory: stage the rows
        local = local.cpu()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = torch.zeros(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, n)
    sizes = sizes.cpu().tolist()
    width, mx = local.shape[1], max(max(sizes), 1)
    padded = torch.zeros((mx, width), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0).to(home)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def _reduce_device(device):
    return "cpu" if dist.get_backend() == "gloo" else device


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value, device):
    """[value of rank 0, value of rank 1, ...] on every rank (one all_gather of a float64)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    out = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return [float(x) for x in out.cpu().tolist()]


def ranks_per_device():
    """How many ranks of this job share one GPU (1 on a real multi-GPU node; > 1 when a gloo test job runs N ranks on a
    1-GPU box): memory budgets (leaf caches) are divided by it."""
    _, _, world = env_world()
    return max(1, -(-world // max(1, torch.cuda.device_count())))


def sum_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
