"""N>1 path on CPU: world_size-2 gloo.  Pictures shard p -> rank p mod G with no
data-path collective; torch.distributed only provides the bench contract's barrier
and max-over-ranks timing, and the gather that puts results back in POC order."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_picture_shard_partition():
    from wrenc_amd import sharding
    for n in (0, 1, 7, 240):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                mine = sharding.picture_shard(n, r, world)
                assert all(sharding.owner_of(p, world) == r for p in mine)
                seen += mine
            assert sorted(seen) == list(range(n))
    with pytest.raises(ValueError):
        sharding.picture_shard(4, 2, 2)


def test_merge_in_poc_order():
    from wrenc_amd import sharding
    assert sharding.merge_in_poc_order([{0: "a", 2: "c"}, {1: "b"}]) == ["a", "b", "c"]
    with pytest.raises(ValueError):
        sharding.merge_in_poc_order([{0: "a"}, {0: "b"}])
    with pytest.raises(ValueError):
        sharding.merge_in_poc_order([{0: "a"}, {2: "b"}])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import hashlib
    from wrenc_amd import sharding, synth
    from oracle import pyoracle as po      # the checker stands in for the per-picture encode
    grp = sharding.Group(backend="gloo")
    n_pic, w, h, qp, depth = 5, 64, 32, 32, 1
    mine = {}
    grp.barrier()
    for poc in sharding.picture_shard(n_pic, grp.rank, grp.world_size):
        y, cb, cr = synth.synth_textured_frame(w, h, poc)
        out = po.encode_picture(y, cb, cr, qp, depth)
        mine[poc] = hashlib.sha256(out["lev_y"].tobytes() + out["rec_y"].tobytes()).hexdigest()
    grp.barrier()
    step_time = grp.max(1.0 + rank)          # max over ranks
    total = grp.sum(len(mine))
    merged = sharding.merge_in_poc_order(grp.gather_objects(mine))
    grp.close()
    q.put((rank, step_time, total, merged))


def test_two_rank_gloo_job_matches_single_process():
    import torch.multiprocessing as mp
    import hashlib
    from wrenc_amd import synth
    from oracle import pyoracle as po
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = []
    for poc in range(5):
        y, cb, cr = synth.synth_textured_frame(64, 32, poc)
        out = po.encode_picture(y, cb, cr, 32, 1)
        want.append(hashlib.sha256(out["lev_y"].tobytes() + out["rec_y"].tobytes()).hexdigest())
    for rank, step_time, total, merged in results:
        assert step_time == 2.0       # max(1.0, 2.0)
        assert total == 5
        assert merged == want
