"""Worker for tests/test_distributed.py: one rank of the read-routing exchange (pantax_amd.pipeline.route_reads) under gloo.
The device side is replaced by a numpy stand-in with the same methods (the kernels themselves are covered by
tests/test_gpu_route.py); what is tested here is the exchange: sizes by one all-to-all, messages by one all-to-all(v),
reassembly in source-rank order."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def slice_messages(rank, world, seed=5):
    """The messages rank `rank` would pack for every owner: random reads in the layout of include/pantax_hip.h
    (n_steps | pstart | pend | qlen | mapq | node ids).  Deterministic, so any process can restate any rank's messages."""
    rng = np.random.default_rng(seed * 100 + rank)
    msgs, nr, nt = [], [], []
    for d in range(world):
        n = int(rng.integers(0, 40)) if (rank + d) % 4 != 3 else 0          # some pairs exchange nothing
        ns = rng.integers(0, 9, size=n).astype(np.uint32)
        cols = [ns] + [rng.integers(0, 1 << 20, size=n).astype(np.uint32) for _ in range(4)]
        ids = rng.integers(1, 1 << 30, size=int(ns.sum())).astype(np.uint32)
        msgs.append(np.concatenate(cols + [ids]).astype(np.uint32))
        nr.append(n)
        nt.append(int(ns.sum()))
    return msgs, np.array(nr, dtype=np.uint64), np.array(nt, dtype=np.uint64)


class FakeEngine:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.received = None

    def route_pack(self, owner, world):
        self.msgs, nr, nt = slice_messages(self.rank, world)
        return "route", nr, nt

    def route_messages(self, rt, world):
        return self.msgs

    def route_buffer(self, rt, world, on_device=False):
        raise AssertionError("the gloo path moves host buffers")

    def route_free(self, rt):
        pass

    def reads_from_routed(self, recv, nr_from, nt_from, on_device=False):
        assert not on_device
        self.received = (np.asarray(recv, dtype=np.uint32).copy(), np.asarray(nr_from).copy(), np.asarray(nt_from).copy())


if __name__ == "__main__":
    import torch.distributed as dist
    from pantax_amd.pipeline import TorchComm, route_reads
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = TorchComm(device=None)
    eng = FakeEngine(rank, world)
    stats = route_reads(eng, None, comm)
    recv, nr_from, nt_from = eng.received
    tot = comm.allreduce_sum(np.array([stats["sent_reads"], stats["recv_reads"], 1.0]))
    with open("%s.%d" % (out, rank), "w") as f:
        json.dump(dict(recv=recv.tolist(), nr=[int(x) for x in nr_from], nt=[int(x) for x in nt_from], stats=stats, tot=tot.tolist()), f)
    dist.barrier()
    dist.destroy_process_group()
