"""Picture sharding across the GPUs of one node.

Every picture is an independent IDR I-slice (main.rs:296,358; CABAC is
re-initialised per picture, ctu_encoder.rs:38-47), so pictures shard with NO
data-path collective: picture p goes to rank p mod G and each rank runs its own
context on its own HIP stream.  torch.distributed is used only for the bench
contract's barrier and max-over-ranks timing (backend "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests).
"""
import os


def world_from_env():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def picture_shard(num_pictures, rank, world_size):
    """POCs owned by `rank`: p mod world_size == rank, ascending."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, num_pictures, world_size))


def owner_of(poc, world_size):
    return poc % world_size


class Group:
    """Thin wrapper: no-ops in a single process, torch.distributed otherwise."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world_size = world_from_env()
        self.device = device
        self._dist = None
        if self.world_size > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend=backend or "nccl", rank=self.rank, world_size=self.world_size)
            self._dist = dist

    def _tensor(self, values, dtype):
        import torch
        dev = self.device if self.device is not None else "cpu"
        return torch.tensor(values, dtype=dtype, device=dev)

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value):
        """Max of a float over all ranks (the bench contract's step time)."""
        if self._dist is None:
            return float(value)
        import torch
        t = self._tensor([float(value)], torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = self._tensor([float(value)], torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def gather_objects(self, obj):
        """Every rank's object on every rank, in rank order (used to merge per-picture
        results back into POC order on the host that writes the bitstream)."""
        if self._dist is None:
            return [obj]
        out = [None] * self.world_size
        self._dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def merge_in_poc_order(per_rank_results):
    """per_rank_results: list (rank order) of dicts poc -> result.  Returns the
    results as one list in POC order; raises if a POC is missing or duplicated."""
    merged = {}
    for d in per_rank_results:
        for poc, r in d.items():
            if poc in merged:
                raise ValueError("POC %d produced twice" % poc)
            merged[poc] = r
    n = len(merged)
    missing = [p for p in range(n) if p not in merged]
    if missing:
        raise ValueError("missing POCs %s" % missing[:8])
    return [merged[p] for p in range(n)]
