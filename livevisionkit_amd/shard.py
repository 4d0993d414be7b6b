"""Stream sharding across the GPUs of one node: independent video streams, one per device, no data-path collective
(SURVEY.md section 8e).  Only the timing barrier / reductions go through torch.distributed (RCCL on GPUs, gloo on CPU)."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def streams_for_rank(n_streams, rank, world):
    """stream i -> rank i mod world (round robin); returns the stream ids this rank owns."""
    return [i for i in range(n_streams) if i % world == rank]


def reduce_timing(elapsed_s, units, device=None):
    """(max elapsed over ranks, total units over ranks). Falls back to the local values without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(units)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def gather_rank_reports(report):
    """Every rank's small report dict (stream seed, device, frames, seconds, NUMA CPUs ...) on every rank, in rank order.  Bookkeeping
    for the bench line only -- a gloo object gather, nothing of the data path."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [report]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, report)
    return out


class LegBarriers:
    """Barrier bookkeeping for legs that every rank runs at the same time (bench.py --gpus N > 1: the host-fed leg, configs 4 / 5): each leg meets the
    other ranks at a fixed number of barriers; a leg that FAILS half-way on one rank must still meet them, or the ranks that did not fail wait for
    it for ever.  barrier() counts, settle(n) makes up the calls a failed leg skipped."""

    def __init__(self, barrier):
        self._barrier, self.calls = barrier, 0

    def __call__(self):
        self.calls += 1
        self._barrier()

    def settle(self, target):
        while self.calls < target:
            self()


def run_legs(legs, barrier, per_leg=2):
    """legs: [(key, callable(barrier))], each callable calling `barrier` exactly `per_leg` times when it succeeds.  Returns {key: result or
    {"error": repr}}; whatever a leg did, this rank has passed per_leg * len(legs) barriers when it returns."""
    b = LegBarriers(barrier)
    out = {}
    for k, (key, fn) in enumerate(legs):
        try:
            out[key] = fn(b)
        except Exception as e:          # an extra leg must never break the contract line -- nor hang the other ranks
            out[key] = {"error": repr(e)}
        b.settle(per_leg * (k + 1))
    return out


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(device_index, sysfs="/sys"):
    """CPUs of the NUMA node the GPU hangs off (the host thread that drives a stream, and the pinned buffers it first touches,
    belong next to that GPU: SURVEY.md section 8e).  Returns [] when the topology cannot be read."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read())
        if node < 0:
            return []
        return _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read())
    except Exception:
        return []


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node the GPU hangs off, or -1 when the topology cannot be read (single-node hosts report -1 or 0)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read())
    except Exception:
        return -1


def numa_node_of_address(address, numa_maps="/proc/self/numa_maps", maps="/proc/self/maps"):
    """NUMA node holding the pages of the mapping that CONTAINS `address` (the node with the most pages of that mapping), or -1 -- also when the
    address lies in no mapping, or in one the kernel lists without per-node page counts (device-file backed pinned memory): the mapping is
    looked up in /proc/self/maps, which has its end address (numa_maps has only the starts; the closest start below an address in a gap is
    somebody else's mapping).  For checking where pinned frame planes landed: eight host-fed 4K streams move ~80 GB/s per GPU through host DRAM
     -- on the socket the GPU hangs off, or across the inter-socket links."""
    try:
        start = None
        for line in open(maps):
            lo, _, hi = line.split()[0].partition("-")
            lo, hi = int(lo, 16), int(hi, 16)
            if lo <= address < hi:
                start = lo
                break
        if start is None:
            return -1
        for line in open(numa_maps):
            parts = line.split()
            if int(parts[0], 16) != start:
                continue
            pages = {int(k[1:]): int(v) for k, v in (t.split("=") for t in parts[1:] if t.startswith("N") and "=" in t and t[1:].split("=")[0].isdigit())}
            return max(pages, key=pages.get) if pages else -1
        return -1
    except Exception:
        return -1


def bind_to_gpu_numa(device_index):
    """Pins the calling process to the CPUs local to its GPU.  Returns the CPU list used ([] = left unbound)."""
    cpus = gpu_numa_cpus(device_index)
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0))) if cpus else []
    if allowed:
        os.sched_setaffinity(0, allowed)
    return allowed
