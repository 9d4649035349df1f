"""The N > 1 path on CPU: two gloo ranks shard the rows of one matrix, gather
them on rank 0, and the result equals the single-process matrix.  The rows
themselves come from the oracle here (no GPU in this suite); on the GPU box
bench.py fills them with andi_hip_scan_rows and gathers over RCCL with the same
code (andi_amd/shard.py)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, seqs, out_path):
    import torch
    import torch.distributed as dist
    from andi_amd import shard
    from oracle import orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = len(seqs)
    r0, r1 = shard.row_block(total, world, rank)
    block = torch.zeros((shard.max_rows(total, world), total, 17), dtype=torch.int32)
    for i in range(r0, r1):  # dist_hack.h:46-68 for the owned subjects only
        E = orc.OracleEsa(seqs[i])
        for j in range(total):
            if i == j:
                row = np.zeros(17, np.uint32)
                row[0] = row[16] = 9
            else:
                row = E.dist_anchor(seqs[j])
            block[i - r0, j] = torch.from_numpy(row.view(np.int32))
        E.close()
    full = shard.gather_matrix(block, total, dist, world, rank)
    if rank == 0:
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


def test_row_blocks_partition():
    from andi_amd import shard
    for total in (1, 2, 3, 29, 42, 60, 80, 3085):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard.row_block(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == shard.max_rows(total, world)
    assert [shard.weak_scaling_set_size(n) for n in (1, 2, 4, 8)] == [29, 42, 60, 80]


@pytest.mark.timeout(300)
def test_two_ranks_gather_equals_single_process(tmp_path, orc):
    import torch.multiprocessing as mp
    from andi_amd import synth
    seqs, _ = synth.genome_set(5, 20000, 0.005, 0.05, seed=77)
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(2, _free_port(), seqs, out), nprocs=2, join=True)
    got = np.load(out)
    want = orc.dist_matrix(seqs, threads=2)
    assert got.shape == want.shape and (got == want).all()
