"""The seam's tiling of the subject rows over the GPUs of a node (api.hip: row_block, the parallel loop of
src/dist_hack.h:46-47) against the one the one-process-per-GPU entry uses (andi_amd/shard.py): the same contiguous blocks for
every n = 1 ... 3085 (BASELINE's config 3) and 1 ... 8 parts -- through the C-ABI (andi_hip_row_block), no GPU touched."""
import ctypes as C

from andi_amd import lib, shard


def test_row_partition_of_the_seam_equals_the_shard_helper():
    L = lib.load()
    f, l = C.c_size_t(0), C.c_size_t(0)
    for parts in range(1, 9):
        for n in list(range(1, 400)) + [1000, 2047, 2048, 3084, 3085, 3086, 10 ** 6 + 7]:
            prev = 0
            sizes = []
            for k in range(parts):
                L.andi_hip_row_block(n, parts, k, C.byref(f), C.byref(l))
                assert (f.value, l.value) == shard.row_block(n, parts, k), (n, parts, k)
                assert f.value == prev and l.value >= f.value  # contiguous, in order
                prev = l.value
                sizes.append(l.value - f.value)
            assert prev == n and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # every n up to BASELINE's 3085 at 8 parts (the north_star's target), once
    for n in range(1, 3086):
        assert [lib.row_block(n, 8, k) for k in range(8)] == [shard.row_block(n, 8, k) for k in range(8)]
    # a part that does not exist owns nothing
    L.andi_hip_row_block(10, 4, 4, C.byref(f), C.byref(l))
    assert (f.value, l.value) == (0, 0)
    L.andi_hip_row_block(10, 0, 0, C.byref(f), C.byref(l))
    assert (f.value, l.value) == (0, 0)
