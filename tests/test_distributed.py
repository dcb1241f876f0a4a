"""Region sharding + the one exchange step (VCF gather), rehearsed with gloo on CPU at world_size 2."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from focalsv_amd import pipeline


def test_shard_regions_balanced_and_complete():
    work = [5, 100, 7, 50, 50, 3, 90, 1]
    shards = pipeline.shard_regions(work, 3)
    assert sorted(i for s in shards for i in s) == list(range(len(work)))
    loads = [sum(work[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(work)
    assert pipeline.shard_regions(work, 3) == shards  # deterministic
    assert pipeline.shard_regions(work, 1) == [list(range(len(work)))]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lines = {0: ["chr21\t500\ta\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n", "chr2\t9\tb\tAT\tA\t20\tPASS\tSVLEN=-1;SVTYPE=DEL\tGT\t1/1\n"],
             1: ["chr21\t20\tc\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n"]}[rank]
    if rank == 1 and world == 2:
        pass
    out = pipeline.gather_vcf(lines)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_vcf_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    assert [l.split('\t')[2] for l in res[0]] == ["b", "c", "a"]  # chr2 before chr21, then by position


def test_gather_vcf_single_process_sorts():
    lines = ["chr21\t9\tx\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n", "chr21\t3\ty\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n"]
    assert [l.split('\t')[2] for l in pipeline.gather_vcf(lines)] == ["y", "x"]
