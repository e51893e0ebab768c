"""
Multi-GPU decode: data-parallel over codewords, one process per GPU.

Codewords are independent and the graph / weight tables / quantiser LUTs are a few KB,
so every rank holds a full decoder and decodes its own contiguous slice of the batch;
nothing is exchanged during the iterations.  The only collective is ONE all-gather of
the hard decisions at the end (``torch.distributed.all_gather_into_tensor`` -- RCCL over
xGMI with backend "nccl", gloo in the CPU tests).  Wire format: bit-packed
``uint8[B, ceil(n/8)]`` written directly by the engine's output kernel (1/32 of the
int32 bits the single-GPU API returns), so a 32768 x 16200 shard is 66 MB per rank and
each peer's shard travels its own direct xGMI link.

The reference has no distributed code at all; this is the build's own
extension behind the same decoder objects.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """contiguous [begin, end) slice of `total` codewords owned by `rank`; the first
    total % world_size ranks get one extra codeword"""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside world")
    base, extra = divmod(int(total), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pack_bits(bits: torch.Tensor) -> torch.Tensor:
    """int bits [B, n] -> uint8 [B, ceil(n/8)], bit j at byte j//8, position j%8
    (same format the engine's output kernel writes)"""
    B, n = bits.shape
    nb = (n + 7) // 8
    padded = torch.zeros((B, nb * 8), dtype=torch.uint8, device=bits.device)
    padded[:, :n] = bits.to(torch.uint8)
    weights = (2 ** torch.arange(8, device=bits.device, dtype=torch.int32)).to(torch.uint8)
    return (padded.view(B, nb, 8) * weights).sum(dim=2).to(torch.uint8)


def unpack_bits(packed: torch.Tensor, n: int) -> torch.Tensor:
    """uint8 [B, ceil(n/8)] -> int32 [B, n]"""
    B = packed.shape[0]
    shifts = torch.arange(8, device=packed.device, dtype=torch.uint8)
    bits = (packed.unsqueeze(-1) >> shifts) & 1
    return bits.reshape(B, -1)[:, :n].to(torch.int32)


def all_gather_hard_decisions(packed_local: torch.Tensor, total: int,
                              group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Gather every rank's packed hard decisions into the full [total, nbytes] array, in
    global codeword order.  Shards follow ``shard_range``; ragged shards are padded to the
    largest one for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if packed_local.shape[0] != total:
            raise ValueError("single-process gather: local shard must be the whole batch")
        return packed_local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, e = shard_range(total, world, rank)
    if packed_local.shape[0] != e - b:
        raise ValueError(f"rank {rank}: shard has {packed_local.shape[0]} codewords, expected {e - b}")
    nbytes = packed_local.shape[1]
    per = -(-total // world)                       # largest shard
    send = packed_local
    if send.shape[0] != per:
        send = torch.zeros((per, nbytes), dtype=torch.uint8, device=packed_local.device)
        send[: e - b] = packed_local
    send = send.contiguous()
    recv = torch.empty((world * per, nbytes), dtype=torch.uint8, device=packed_local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    if total == world * per:
        return recv
    out = torch.empty((total, nbytes), dtype=torch.uint8, device=packed_local.device)
    for r in range(world):
        rb, re_ = shard_range(total, world, r)
        out[rb:re_] = recv[r * per: r * per + (re_ - rb)]
    return out


def decode_sharded(engine, llr_local: torch.Tensor, total: int, *, early_stop: bool = True,
                   group: Optional[dist.ProcessGroup] = None):
    """Decode this rank's slice on its GPU and all-gather the packed hard decisions.
    Returns (packed_all uint8[total, ceil(n/8)], local DecodeResult)."""
    res = engine.decode(llr_local, early_stop=early_stop, want_bits=False, want_posterior=False,
                        want_packed=True)
    return all_gather_hard_decisions(res.packed_bits, total, group), res


def all_reduce_gradients(parameters, group: Optional[dist.ProcessGroup] = None, average: bool = True) -> int:
    """Data-parallel training: sum (or average) the `.grad` of every parameter over the ranks with ONE
    all-reduce of a flat buffer (the decoders have a few hundred scalar weights; one bucket, not one collective
    per [1]-shaped parameter).  Parameters without a gradient on this rank contribute zeros, so every rank ends
    with the same gradients.  Returns the number of scalars reduced.  RCCL with backend "nccl" when the
    parameters live on the GPU, gloo on the CPU."""
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return 0
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sum(p.numel() for p in params)
    dev, dt = params[0].device, params[0].dtype
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(device=dev, dtype=dt)
                      for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        k = p.numel()
        g = flat[off:off + k].reshape(p.shape).to(device=p.device, dtype=p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += k
    return off
