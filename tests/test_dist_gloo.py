"""N>1 path on CPU (gloo, world_size 2): the constant-table blob is built on rank 0 only, broadcast,
validated by checksum on every rank, and the burst sharding covers every burst exactly once with no
data-path collective.  (The GPU ranks do exactly this over RCCL in bench.py.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import _pkg
    pkg = _pkg.load()
    from openbts_ttsou_amd import dist as tdist
    r, w = tdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # only rank 0 may build: poison the builder elsewhere
    if rank != 0:
        pkg.build_tables_host = lambda sps: (_ for _ in ()).throw(AssertionError("non-root rank built tables"))
    blob, t = tdist.broadcast_tables(pkg, 4, device=None, src=0)
    assert pkg.tables_valid(blob)
    lo, hi = tdist.shard_range(total, rank, world)
    tmax = tdist.max_over_ranks(1.0 + rank)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), blob=blob, lo=lo, hi=hi, tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [65536, 1001])
def test_broadcast_and_sharding_world2(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import _pkg
    pkg = _pkg.load()
    ref = pkg.build_tables_host(4)
    rs = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    for r in rs:
        assert np.array_equal(r["blob"], ref)
        assert float(r["tmax"]) == 2.0
    spans = sorted((int(r["lo"]), int(r["hi"])) for r in rs)
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_corrupted_blob_is_rejected():
    sys.path.insert(0, ROOT)
    import _pkg
    pkg = _pkg.load()
    blob = pkg.build_tables_host(4).copy()
    assert pkg.tables_valid(blob)
    blob[5000] ^= 1
    assert not pkg.tables_valid(blob)


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    import _pkg
    _pkg.load()
    from openbts_ttsou_amd.dist import shard_range
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
