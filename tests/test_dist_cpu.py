"""CPU (gloo, world_size 2): the control plane of the sharded pricing path — mailbox creation and
name broadcast, block partition, and the mailbox protocol itself (post own slot, poll peers, reduce
with the device's tie-break rule) emulated on the mapped shared-memory object the kernels use."""
import os
import struct
import sys
import time

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from minilp_amd import dist as md


def test_shard_ranges_partition_positions():
    for n in (1, 7, 100000, 100003):
        for world in (1, 2, 3, 8):
            rs = [md.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1


def test_combine_candidates_tie_break():
    assert md.combine_candidates([(1.0, 5), (2.0, 9), (2.0, 3), (0.5, -1)]) == (2.0, 3)   # score desc, position asc
    assert md.combine_candidates([(-1.0, -1), (-1.0, -1)])[1] == -1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    box = [md.create_mailbox(world) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    name = box[0]
    path = "/dev/shm" + name
    assert os.path.getsize(path) == md.mailbox_bytes(world)
    mm = np.memmap(path, dtype=np.uint8, mode="r+")
    rec = md.MAILREC_BYTES

    def slot(kind, epoch, r):
        return (kind * 2 + (epoch & 1)) * world * rec + r * rec

    results = []
    for epoch in range(1, 6):  # five pricing exchanges with the device protocol
        score, pos = float((rank * 7 + epoch * 3) % 5), rank * 10 + epoch
        off = slot(0, epoch, rank)
        mm[off + 8: off + 24] = np.frombuffer(struct.pack("<dd", score, float(pos)), dtype=np.uint8)  # payload first
        mm[off: off + 8] = np.frombuffer(struct.pack("<Q", epoch), dtype=np.uint8)                      # epoch last
        cands = []
        for r2 in range(world):
            o2 = slot(0, epoch, r2)
            t0 = time.time()
            while struct.unpack("<Q", bytes(mm[o2: o2 + 8]))[0] != epoch:
                assert time.time() - t0 < 20
            sc, ps = struct.unpack("<dd", bytes(mm[o2 + 8: o2 + 24]))
            cands.append((sc, int(ps)))
        results.append(md.combine_candidates(cands))
    # the dual loop's exchanges on the same object: kind 2 = all-reduce MIN of the pass-1 ratio, kind 3 = pass-2
    # candidate with two payloads (alpha_rq, d_q); kinds are disjoint slots, so they may share epochs with kind 0
    for epoch in range(1, 4):
        mine = float((rank + 2) * epoch) / 7.0
        off = slot(2, epoch, rank)
        mm[off + 8: off + 16] = np.frombuffer(struct.pack("<d", mine), dtype=np.uint8)
        mm[off: off + 8] = np.frombuffer(struct.pack("<Q", epoch), dtype=np.uint8)
        vals = []
        for r2 in range(world):
            o2 = slot(2, epoch, r2)
            t0 = time.time()
            while struct.unpack("<Q", bytes(mm[o2: o2 + 8]))[0] != epoch:
                assert time.time() - t0 < 20
            vals.append(struct.unpack("<d", bytes(mm[o2 + 8: o2 + 16]))[0])
        results.append(("min", min(vals)))
        assert min(vals) == 2.0 * epoch / 7.0  # rank 0 holds the smallest ratio
    assert slot(5, 1, world - 1) + rec <= md.host_box_bytes(world) <= md.mailbox_bytes(world) and md.KINDS == 6
    gathered = [None] * world
    dist.all_gather_object(gathered, results)
    if rank == 0:
        q.put(all(g == gathered[0] for g in gathered) and len(results) == 8)
        md.remove_mailbox(name)
    dist.barrier()
    dist.destroy_process_group()


def test_mailbox_protocol_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29641, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _bench_cli(args, env_extra):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MLP_OVERSUBSCRIBE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)


def test_bench_gpus_flag_is_read_and_a_mismatched_world_is_refused():
    """`--gpus N` is the number of ranks that run, not a label: (i) N > 1 without a launcher makes bench.py start N ranks itself,
    which on a box with fewer GPUs (here: none) is refused with rc 2 unless MLP_OVERSUBSCRIBE=1; (ii) a launcher whose WORLD_SIZE
    differs from --gpus is refused as well.  Neither prints a JSON line."""
    r = _bench_cli(["--gpus", "2", "--steps", "20", "--warmup", "5"], {})
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout
    r = _bench_cli(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and "{" not in r.stdout
    r = _bench_cli(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "{" not in r.stdout
