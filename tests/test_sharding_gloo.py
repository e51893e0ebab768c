"""
Multi-process tests of the N > 1 path on CPU (gloo, world_size 2 and 3): contiguous
shards, ragged batches, the single all-gather of bit-packed hard decisions, global order.
The decode itself needs a GPU; here every rank's "decode" is a deterministic stand-in so
that the collective plumbing bench.py and decode_sharded() use is what is under test.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_bits(total, n):
    rng = np.random.default_rng(42)
    return rng.integers(0, 2, (total, n)).astype(np.int32)


def _worker(rank, world, port, total, n, q):
    import sys
    from conftest import PKG  # noqa: F401  (sys.path set up by conftest import)
    import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bits = torch.from_numpy(_fake_bits(total, n))
        b, e = sharding.shard_range(total, world, rank)
        local = sharding.pack_bits(bits[b:e])                      # what the engine's output kernel writes
        gathered = sharding.all_gather_hard_decisions(local, total)
        ok = torch.equal(sharding.unpack_bits(gathered, n), bits)
        # every rank must hold the same, complete result
        chk = torch.tensor([int(gathered.to(torch.int64).sum())])
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        ok = ok and all(int(x) == int(chk) for x in lst)
        q.put((rank, bool(ok), tuple(gathered.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,n", [(2, 64, 96), (2, 37, 1998), (3, 10, 7)])
def test_all_gather_of_packed_hard_decisions(world, total, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, f"rank {rank} gathered wrong data"
        assert shape == (total, (n + 7) // 8)


def test_shard_ranges_partition_the_batch():
    import sharding
    for total in (0, 1, 7, 64, 65536, 262144 + 5):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def test_pack_unpack_roundtrip_and_format():
    import sharding
    bits = torch.from_numpy(_fake_bits(33, 1998))
    packed = sharding.pack_bits(bits)
    assert packed.dtype == torch.uint8 and packed.shape == (33, 250)
    assert torch.equal(sharding.unpack_bits(packed, 1998), bits)
    # bit j of a codeword lives at byte j // 8, position j % 8 (include/ldpc_hip.h)
    one = torch.zeros((1, 20), dtype=torch.int32)
    one[0, 10] = 1
    assert sharding.pack_bits(one).tolist() == [[0, 4, 0]]
    # single process: the "gather" is the identity
    assert sharding.all_gather_hard_decisions(packed, 33) is packed


def _grad_worker(rank, world, port, q):
    from conftest import PKG  # noqa: F401
    import sharding
    from ldpc_decoder import create_test_ldpc_code
    from neural_2d_decoder import Neural2DMinSumDecoder
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = Neural2DMinSumDecoder(create_test_ldpc_code(), 2, 3)      # same init on every rank
        names = [k for k, _ in model.named_parameters()]
        for i, (k, p) in enumerate(model.named_parameters()):            # rank-dependent fake gradients, one left at None
            if not (rank == 1 and i == 2):
                p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
        count = sharding.all_reduce_gradients(model.parameters())
        want = []
        for i in range(len(names)):
            tot = sum(0.0 if (r == 1 and i == 2) else float((r + 1) * (i + 1)) for r in range(world))
            want.append(tot / world)
        got = [float(p.grad.item()) for p in model.parameters()]
        q.put((rank, count == len(names), got, want))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_all_reduce():
    """one bucketed all-reduce averages the table gradients over the ranks (gloo here, RCCL on the GPUs)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, got, want in res:
        assert ok
        assert got == pytest.approx(want)
    assert res[0][2] == res[1][2]
