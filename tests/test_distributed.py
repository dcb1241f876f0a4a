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


def _queue_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    rng = random.Random(4)
    work = [rng.choice([14_000, 26_000, 50_000, 1_100_000]) for _ in range(1000)]   # widths like the whole-genome BED (config 4)
    rq = pipeline.RegionQueue(work, batch=16)
    got = []
    for b in rq.batches():
        got += b
        if rank == 1:
            import time
            time.sleep(0.002)          # a slow rank: the other one should drain most of the tail
    q.put((rank, got, len(rq.static)))
    dist.barrier()
    # a second queue on the same process group (the next chromosome): its cursor starts from zero again
    rq2 = pipeline.RegionQueue(work[:300], batch=16)
    got2 = [i for b in rq2.batches() for i in b]
    q.put((rank + 10, got2, rq2.n_stolen_batches))
    dist.barrier()
    dist.destroy_process_group()


def test_region_queue_gloo_world2_covers_every_region_once():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_queue_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(4)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    second = [r for r in res if r[0] >= 10]
    res = [r for r in res if r[0] < 10]
    assert sorted(i for _, got, _ in second for i in got) == list(range(300))      # second queue: every region once, tail included
    assert sum(n for _, _, n in second) == (300 - int(300 * 0.75) + 15) // 16
    allr = sorted(i for _, got, _ in res for i in got)
    assert allr == list(range(1000))                       # every region exactly once
    by = {r: (len(got), ns) for r, got, ns in res}
    assert by[0][0] - by[0][1] > by[1][0] - by[1][1]      # the fast rank took more of the shared tail


def test_region_queue_single_process():
    rq = pipeline.RegionQueue([3, 1, 2, 9, 9], batch=2)
    assert sorted(i for b in rq.batches() for i in b) == [0, 1, 2, 3, 4]


def _bed_worker(rank, world, port, q):
    """the bed workload's control flow around a stand-in for the GPU call: RegionQueue -> run_stream lanes -> gather_vcf"""
    import gzip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = [l.split() for l in gzip.open(os.path.join(root, "tests", "golden", "bed_whole_genome.bed.gz"), "rt")]
    rows = [(c, int(a), int(b)) for c, a, b in rows][::20]                 # every 20th line of the whole-genome BED: 1 342 regions, 14 kb .. 1.1 Mb
    work = [b - a + 30000 for _, a, b in rows]

    class Pending:
        def __init__(self, lines):
            self.lines = lines

        def finish(self):
            r = type("R", (), {})()
            r.lines = self.lines
            return r

    def launch(ctx, batch, **kw):        # "calls" one SV at the start of every region of the batch
        import time
        time.sleep(0.0005 * len(batch) * (3 if rank == 1 else 1))           # rank 1 is the slow one
        return Pending(["%s\t%d\tr%d\tA\tAT\t20\tPASS\tSVLEN=1;SVTYPE=INS\tGT\t0/1\n" % (rows[i][0], rows[i][1], i) for i in batch])

    pipeline.launch_hot_path = launch
    rq = pipeline.RegionQueue(work, batch=16)
    lines = []
    dist.barrier()                       # both ranks start on the queue together (a rank that comes up late would find the tail gone)
    pipeline.run_stream(["lane0", "lane1"], rq.batches(), on_result=lambda i, r: lines.extend(r.lines), keep_results=False)
    out = pipeline.gather_vcf(lines)
    q.put((rank, len(lines), rq.n_static_batches, rq.n_stolen_batches, [l.split('\t')[2] for l in out] if rank == 0 else None,
           [(l.split('\t')[0], int(l.split('\t')[1])) for l in out] if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_bed_workload_control_flow_gloo_world2():
    """bench.py --workload bed without the GPU: the uneven whole-genome region set over two ranks -- every region called exactly
    once, the gathered VCF in (chromosome, position) order, the fast rank taking more of the shared tail"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda x: x[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    ids, keys = res[0][4], res[0][5]
    assert sorted(ids) == sorted("r%d" % i for i in range(1342))
    num = lambda c: int(c[3:])
    assert keys == sorted(keys, key=lambda k: (num(k[0]), k[1]))
    assert res[0][1] + res[1][1] == 1342
    assert res[0][3] >= res[1][3]           # stolen batches: the fast rank drained at least as much of the tail (a loaded test machine blurs it)
