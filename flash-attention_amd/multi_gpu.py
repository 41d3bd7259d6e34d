"""Multi-GPU drivers for the forward-attention path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI) as plumbing.

The reference has no multi-device code (README.md:47 lists it as future work), so
nothing here mirrors a reference file; the two modes follow from the math:

* Head sharding.  Every (batch, head) pair is an independent attention problem
  (the reference kernel's grid axes blockIdx.y / blockIdx.z,
  tensor_flash_attention.cu:30-31) and one head is one contiguous L*D slab of each
  tensor, so a rank's shard is a contiguous range of the flattened [B*H] axis and the
  data path needs no collective at all.

* Sequence-parallel ring.  Rank r owns rows [r*L/P, (r+1)*L/P) of Q, K, V and O for
  all heads.  In P steps each K/V block visits every rank once: at step s rank r holds
  block (r - s) mod P, folds it into its carried (O, m, l) with the blockwise entry
  point, and meanwhile passes it to rank r+1 with a send/recv pair.  The merge rule is
  the online-softmax rescale of testing.cu:91-96 applied across blocks.  Non-causal
  attention makes every step the same amount of work on every rank.

* Causal ring (zig-zag).  With the causal mask of attention_cpu (testing.cu:74-77) a
  contiguous split would leave rank 0 with one block of work and rank P-1 with P.  The
  sequence is cut into 2P chunks instead and rank r owns chunks r and 2P-1-r: in every
  ring step each rank then has exactly two (query chunk, key chunk) pairs that are not
  wholly masked.  Pairs on the diagonal use the mask with absolute positions, pairs below it
  none, pairs above it are skipped.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple


def head_shard(total_heads: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) range of the flattened B*H axis owned by ``rank``.
    The first ``total_heads % world`` ranks get one extra head."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(total_heads, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sharded_forward(q, k, v, rank: int, world: int, attention_fn: Optional[Callable] = None):
    """Run this rank's head shard.  q,k,v: the GLOBAL [B,H,L,D] tensors (or any tensors whose
    flattened B*H axis is global); returns (start, stop, O_shard [1, stop-start, L, D]).
    No communication."""
    B, H, L, D = q.shape
    start, stop = head_shard(B * H, world, rank)
    if attention_fn is None:
        import flash_attention_amd as fa

        attention_fn = fa.attention
    qs = q.reshape(1, B * H, L, D)[:, start:stop].contiguous()
    ks = k.reshape(1, B * H, L, D)[:, start:stop].contiguous()
    vs = v.reshape(1, B * H, L, D)[:, start:stop].contiguous()
    return start, stop, attention_fn(qs, ks, vs)


# Compute units the attention kernels leave free while a ring exchanges K/V.  The D = 128 kernels are persistent: one
# workgroup per compute unit, each holding the unit's whole LDS and register file, so a transport kernel of the
# collective library finds no room until an attention kernel ends and the hand-off would queue behind the compute
# instead of running beside it.  Measured on one MI355X with a transport-shaped kernel (8 workgroups x 512 threads,
# 32 KiB LDS, 134 MB): with 8 units free it runs beside the attention kernel; a ring pass of 8 steps at
# L = 131072 then takes 13.1 ms instead of 12.0 ms (1024 equal items no longer divide by the workgroup count).  The last
# step of a pass exchanges nothing and runs without the reservation.
RING_RESERVED_CUS = 8


class _reserved_cus:
    """with _reserved_cus(k): ...  (no-op for k == 0 or when a caller has already reserved units).

    The reservation is PROCESS-WIDE (fa_set_reserved_cus stores one value per process): while the block runs,
    attention launches from other threads or streams of this process also leave k units free.  On exit the value
    found on entry is put back, unless somebody changed the reservation in between: then theirs stays."""

    def __init__(self, k):
        self.k, self.fa, self.before = k, None, None

    def __enter__(self):
        if self.k > 0:
            import flash_attention_amd as fa

            self.fa = fa
            before = fa.lib().fa_get_reserved_cus()
            if before == 0:
                fa.set_reserved_cus(self.k)
                self.before = before
        return self

    def release(self):
        """Put the reservation back early (the ring's last step: nothing travels beside it, so it gets the whole
        chip: one step in `world` runs without the cost of the uneven rounds)."""
        if self.before is not None and self.fa.lib().fa_get_reserved_cus() == self.k:
            self.fa.set_reserved_cus(self.before)
        self.before = None

    def __exit__(self, *exc):
        self.release()
        return False


def ring_forward(q, k, v, group=None, step_fn: Optional[Callable] = None, state_factory: Optional[Callable] = None,
                 rows_per_rank: Optional[Sequence[int]] = None):
    """Sequence-parallel forward.  q,k,v: this rank's rows, contiguous [B,H,L/P,D].
    Returns this rank's rows of O.

    ``rows_per_rank`` (one entry per rank of the group, the same list on every rank) allows shards of different
    lengths, e.g. a sequence that does not divide by the number of ranks: rank r then holds rows_per_rank[r] rows
    of Q, K and V, and the blocks travel through buffers of the largest shard's size.  Without it every rank
    must hold the same number of rows (nothing is exchanged to find out).

    ``step_fn(q, k_blk, v_blk, state, last)`` folds one K/V block into ``state`` and returns
    the output on the last step; ``state_factory(BH, Lq, D, device)`` creates the carried state.
    Both default to the HIP library (GPU tensors required).  Tests inject CPU stand-ins to
    exercise the communication schedule under the gloo backend.
    """
    import torch
    import torch.distributed as dist

    reserve = 0
    if step_fn is None or state_factory is None:
        import flash_attention_amd as fa

        # the general blockwise entry point: any rows per rank and D of 64, 128 or 256; it hands D = 128 blocks
        # to the tuned kernels (whole tiles, and ragged lengths where those win)
        step_fn = step_fn or fa.attention_block_general
        state_factory = state_factory or fa.BlockState
        reserve = RING_RESERVED_CUS

    # (no process group: a single rank, which folds its own block and is done)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, H, Lq, D = q.shape
    state = state_factory(B * H, Lq, D, q.device)
    if world == 1:
        return step_fn(q, k, v, state, True)

    nxt = dist.get_global_rank(group, (rank + 1) % world) if group is not None else (rank + 1) % world
    prv = dist.get_global_rank(group, (rank - 1) % world) if group is not None else (rank - 1) % world

    # K and V travel as one message: [2, B, H, Lk, D], a view of the head of a flat buffer that holds the largest
    # shard (with equal shards: the whole buffer)
    Lk = k.shape[2]
    if rows_per_rank is None:
        rows = [Lk] * world
    else:
        rows = [int(n) for n in rows_per_rank]
        if len(rows) != world or min(rows) < 1 or rows[rank] != Lk or Lq != Lk:
            raise ValueError(f"rows_per_rank {rows} does not describe rank {rank} of {world} holding {Lk} rows")
    per_row = 2 * B * H * D
    bufs = [torch.empty(per_row * max(rows), dtype=k.dtype, device=k.device) for _ in range(2)]
    block = lambda buf, n: buf[:2 * B * H * n * D].view(2, B, H, n, D)
    cur = block(bufs[0], Lk)
    cur[0].copy_(k)
    cur[1].copy_(v)
    out = None
    with _reserved_cus(reserve) as reserved:
        for s in range(world):
            last = s == world - 1
            reqs = []
            if last:
                reserved.release()
            if not last:
                # issued before the compute of this step, so the transfer can run beside it (the attention kernels
                # leave RING_RESERVED_CUS units free for its kernels); the backend orders it after the previous
                # step's kernel, which was the last reader of the buffer being received into
                nxt_blk = block(bufs[(s + 1) & 1], rows[(rank - s - 1) % world])
                ops = [dist.P2POp(dist.isend, cur, nxt, group), dist.P2POp(dist.irecv, nxt_blk, prv, group)]
                reqs = dist.batch_isend_irecv(ops)
            out = step_fn(q, cur[0], cur[1], state, last)
            for r in reqs:
                r.wait()
            if not last:
                cur = nxt_blk
    return out


def ring_block_order(rank: int, world: int):
    """Index of the K/V block rank ``rank`` holds at each step."""
    return [(rank - s) % world for s in range(world)]


def zigzag_chunks(rank: int, world: int) -> Tuple[int, int]:
    """The two chunks (of 2*world) of the sequence that ``rank`` owns in the causal ring."""
    return rank, 2 * world - 1 - rank


def zigzag_schedule(rank: int, world: int):
    """What rank ``rank`` computes in each of the ``world`` ring steps: a list (per step) of
    (q_slot, k_slot, q_chunk, k_chunk, diagonal) with q_slot / k_slot in {0, 1} selecting the
    rank's own and the visiting rank's lower / upper chunk.  Pairs with k_chunk > q_chunk are
    wholly masked and left out; ``diagonal`` pairs (same chunk) need the mask, the others none."""
    mine = zigzag_chunks(rank, world)
    steps = []
    for s in range(world):
        theirs = zigzag_chunks((rank - s) % world, world)
        pairs = []
        for qs, qc in enumerate(mine):
            for ks, kc in enumerate(theirs):
                if kc <= qc:
                    pairs.append((qs, ks, qc, kc, kc == qc))
        steps.append(pairs)
    return steps


def run_zigzag_steps(q2, kv_of_step, rank: int, world: int, chunk_len: int, step_fn: Callable, state_factory: Callable):
    """The compute side of the causal ring for one rank.  q2: [2,B,H,Lc,D] (the rank's lower and
    upper chunk); ``kv_of_step(s)`` returns the visiting [2(k|v), 2(chunk), B,H,Lc,D] tensor of
    step s (and may overlap a transfer with the caller's compute).  Returns [2,B,H,Lc,D].
    ``step_fn(q, k, v, state, last, causal, q_pos0, k_pos0)`` folds one pair."""
    _, B, H, Lc, D = q2.shape
    sched = zigzag_schedule(rank, world)
    states = [state_factory(B * H, Lc, D, q2.device) for _ in range(2)]
    remaining = [sum(1 for st in sched for pr in st if pr[0] == qs) for qs in range(2)]
    outs = [None, None]
    for s in range(world):
        kv = kv_of_step(s)
        for qs, ks, qc, kc, diag in sched[s]:
            remaining[qs] -= 1
            o = step_fn(q2[qs], kv[0][ks], kv[1][ks], states[qs], remaining[qs] == 0, diag, qc * chunk_len, kc * chunk_len)
            if remaining[qs] == 0:
                outs[qs] = o
    import torch

    return torch.stack(outs)


def ring_forward_causal(q2, k2, v2, group=None, step_fn: Optional[Callable] = None,
                        state_factory: Optional[Callable] = None):
    """Causal sequence-parallel forward with zig-zag sharding.  q2, k2, v2: [2,B,H,Lc,D], this rank's
    chunks ``zigzag_chunks(rank, world)`` of a sequence of 2*world*Lc rows.  Returns this rank's rows
    of O in the same layout.  K and V of both chunks travel as one message per step."""
    import torch
    import torch.distributed as dist

    use_gpu_defaults = step_fn is None or state_factory is None
    if step_fn is None or state_factory is None:
        import flash_attention_amd as fa

        def _gpu_step(q, k, v, state, last, causal, q_pos0, k_pos0):
            return fa.attention_block_general(q, k, v, state, last, causal=causal, q_pos0=q_pos0, k_pos0=k_pos0)

        step_fn = step_fn or _gpu_step
        state_factory = state_factory or fa.BlockState

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    reserve = RING_RESERVED_CUS if use_gpu_defaults and world > 1 else 0
    Lc = q2.shape[3]
    nxt = dist.get_global_rank(group, (rank + 1) % world) if group is not None else (rank + 1) % world
    prv = dist.get_global_rank(group, (rank - 1) % world) if group is not None else (rank - 1) % world

    box = {"cur": torch.stack((k2, v2)).contiguous(), "spare": None, "reqs": []}
    if world > 1:
        box["spare"] = torch.empty_like(box["cur"])

    reserved = _reserved_cus(reserve)

    def kv_of_step(s):
        # finish the transfer that brought this step's block, then start the next one: it
        # overlaps this step's compute, and `spare` was last read in the step before
        for r in box["reqs"]:
            r.wait()
        if s + 1 >= world:
            reserved.release()  # the last step exchanges nothing
        if s > 0:
            box["cur"], box["spare"] = box["spare"], box["cur"]
        box["reqs"] = []
        if s + 1 < world:
            ops = [dist.P2POp(dist.isend, box["cur"], nxt, group), dist.P2POp(dist.irecv, box["spare"], prv, group)]
            box["reqs"] = dist.batch_isend_irecv(ops)
        return box["cur"]

    with reserved:
        return run_zigzag_steps(q2, kv_of_step, rank, world, Lc, step_fn, state_factory)
