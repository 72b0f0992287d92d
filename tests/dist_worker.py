"""Worker for tests/test_distributed.py: one rank of the N>1 finalisation path under gloo."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_local(rank, n_species, step=0):
    """Deterministic stand-in for local_stage output of one rank (what the device stages would return)."""
    rng = np.random.default_rng(100 + rank + 1000 * step)
    keep = (rng.random(n_species) < 0.8).astype(np.uint8)
    absolute = np.where(keep == 1, rng.lognormal(1.0, 2.0, n_species), 0.0)
    absolute[0] = 1e-7 if rank == 1 else absolute[0]     # one species under the -a cut
    keep[0] = 1
    rows, s_all, s_pass = [], np.zeros(n_species), np.zeros(n_species)
    h = 0
    for s in range(n_species):
        for _ in range(3):
            cov = float(rng.lognormal(1.0, 1.0))
            if keep[s]:
                s_all[s] += cov
                if rng.random() < 0.7:
                    s_pass[s] += cov
                    rows.append((s, h, cov, 0.9, 1.0, cov, cov, 0.01, 0.05))
            h += 1
    return dict(keep=keep, absolute=absolute, s_all=s_all, s_pass=s_pass, rows=rows, stats={})


def names(rank, n_species):
    return ["sp%d_%d" % (rank, s) for s in range(n_species)], ["hap%d_%d" % (rank, h) for h in range(3 * n_species)]


if __name__ == "__main__":
    import torch.distributed as dist
    from pantax_amd.pipeline import StepConfig, TorchComm, finalize_begin, finalize_end, finalize_stage
    out = sys.argv[1]
    n_species = [4, 6]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    comm = TorchComm(device=None)
    sn, hn = names(rank, n_species[rank])
    species_rows, strain_rows, n_active = finalize_stage(fake_local(rank, n_species[rank]), sn, hn, StepConfig(), comm,
                                                         shard_max=max(n_species), rows_max=3 * max(n_species))
    # a stream of steps with the exchange of step i in flight while step i+1 is prepared (profile_steps_pipelined's order)
    kw = dict(shard_max=max(n_species), rows_max=3 * max(n_species))
    locs = [fake_local(rank, n_species[rank], step) for step in range(1, 5)]
    seq = [finalize_stage(l, sn, hn, StepConfig(), comm, **kw) for l in locs]
    pipe, pending = [], None
    for l in locs:
        nxt = finalize_begin(l, hn, comm, **kw)
        if pending is not None:
            pipe.append(finalize_end(pending, sn, hn, StepConfig(), comm))
        pending = nxt
    pipe.append(finalize_end(pending, sn, hn, StepConfig(), comm))
    assert pipe == seq, "pipelined exchange differs from the sequential one"
    if rank == 0:
        assert len(seq[0][1]) > 0 and seq[0] != seq[1]
        json.dump(dict(species=species_rows, strain=strain_rows), open(out, "w"))
    dist.destroy_process_group()
