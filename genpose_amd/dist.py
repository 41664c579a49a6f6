"""Multi-GPU execution: one process per GPU, clouds sharded across ranks, ONE all-gather of the results.

Clouds are independent units (all K candidates of a cloud stay on one GPU: they share `pts_feat` and are ranked
together), so the encoder, the sampler, the energy network, ranking and aggregation are rank-local and the only
exchange on the path is the final gather: per rank [n_local, K, 9+2] (+ aggregated [n_local, 7]),
a few hundred KB - latency-bound on xGMI, so a single RCCL all_gather per result tensor, no bucketing.
`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests run the same code on "gloo".

Caveat (documented deviation): the reference couples all rows of a batch through the RK45 error norm and the PC
sampler's batch-mean gradient norm; with sharding that coupling becomes shard-local, exactly as
it already is batch-local in the reference (results for a cloud depend on its batch-mates at the 1e-5 level).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced split of n items: the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_ragged(t, n_total, group=None):
    """Gathers the per-rank leading-dim shards produced by `shard_bounds` into one [n_total, ...] tensor on every rank.
    Shards are padded to the largest shard so a single fixed-size all_gather is used."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_len = (n_total + world - 1) // world
    pad = max_len - t.shape[0]
    if pad > 0:
        t = torch.cat([t, t.new_zeros((pad,) + tuple(t.shape[1:]))], dim=0)
    # RCCL (backend 'nccl') gathers device tensors in place; gloo has no all_gather for device tensors (the one-device rehearsal of the
    # N-rank job and the CPU tests run on it): stage through the host there and bring the result back
    home = t.device
    if t.is_cuda and dist.get_backend(group) != "nccl":
        t = t.cpu()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous(), group=group)
    pieces = []
    for r in range(world):
        s, e = shard_bounds(n_total, r, world)
        pieces.append(out[r][: e - s])
    return torch.cat(pieces, dim=0).to(home)


class ShardedInference:
    """Wraps any per-rank inference callable `infer(clouds [n_local,1024,3]) -> dict[str, Tensor]` (leading dim n_local):
    every rank processes its shard of the clouds and receives the full result set."""

    def __init__(self, infer, group=None):
        self.infer, self.group = infer, group

    def __call__(self, clouds):
        n = clouds.shape[0]
        if not (dist.is_available() and dist.is_initialized()):
            return self.infer(clouds)
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        s, e = shard_bounds(n, rank, world)
        local = self.infer(clouds[s:e]) if e > s else None
        # ranks with an empty shard still take part in the collectives: shapes come from a non-empty rank (rank 0)
        meta = [None]
        if rank == 0:
            if local is None:
                raise ValueError("ShardedInference needs at least as many clouds as ranks")
            meta[0] = {k: (tuple(v.shape[1:]), v.dtype) for k, v in local.items()}
        # `src` is a GLOBAL rank: the group's rank 0 (a sub-group need not contain global rank 0)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast_object_list(meta, src=src, group=self.group)
        out = {}
        for k, (shape, dtype) in meta[0].items():
            t = local[k] if local is not None else torch.empty((0,) + shape, dtype=dtype, device=clouds.device)
            out[k] = all_gather_ragged(t, n, self.group)
        return out


# ---------------------------------------------------------------------------------------------- tracking (BASELINE configs[4])
def sequence_owner(seq, n_sequences, world):
    """Rank that owns sequence `seq`: the contiguous balanced split of `shard_bounds` (whole sequences per rank)."""
    for r in range(world):
        s, e = shard_bounds(n_sequences, r, world)
        if s <= seq < e:
            return r
    raise ValueError(f"sequence {seq} outside 0..{n_sequences - 1}")


class ShardedTracking:
    """Tracking over many sequences on N GPUs (runners/evaluation_tracking.py:262-337 runs ONE sequence per process call;
    BASELINE configs[4] streams many).  Frames of a sequence are strictly sequential (warm start from the previous frame) and
    sequences are independent, so the path does not shard below a sequence: every rank owns WHOLE sequences (replicas only, no
    data-path collective) and drives them in lock-step through one `MultiSequenceTracker`-like object, which lets the frames its
    sequences are at share every launch.  The only exchange is the final gather of the per-frame results.

    make_tracker(n_local) -> object with .step(frames) -> list (one dict of tensors or None per local sequence), e.g.
                             `lambda n: MultiSequenceTracker(score_agent, energy_agent, n)`
    """

    def __init__(self, make_tracker, n_sequences, group=None):
        self.group, self.n = group, n_sequences
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.lo, self.hi = shard_bounds(n_sequences, self.rank, self.world)
        self.tracker = make_tracker(self.hi - self.lo) if self.hi > self.lo else None

    def owned(self):
        return range(self.lo, self.hi)

    def run(self, frame_source, keys=("average_sRT",), gather=True):
        """frame_source(seq, t) -> (pts [n,1024,3] device, model_names, gt_RT [n,4,4]) or None once sequence `seq` has ended (it is
        only ever called for sequences this rank owns).  Returns {seq: [per-frame dict of CPU tensors for `keys`]} - for every
        sequence of the job when `gather` (object all-gather of the final results), else for the local ones."""
        local = {s: [] for s in self.owned()}
        t = 0
        live = set(self.owned())
        while live:
            frames = []
            for s in self.owned():
                f = frame_source(s, t) if s in live else None
                if f is None:
                    live.discard(s)
                frames.append(f)
            if not live:
                break
            outs = self.tracker.step(frames)
            for s, o in zip(self.owned(), outs):
                if o is not None:
                    # device-side copies: nothing in the streaming loop waits for the GPU (a .cpu() here is a synchronisation per
                    # sequence and frame); the results come down once, after the last frame
                    local[s].append({k: o[k].detach().clone() for k in keys})
            t += 1
        local = {s: [{k: v.cpu() for k, v in fr.items()} for fr in frs] for s, frs in local.items()}
        if not gather or self.world == 1:
            return local
        parts = [None] * self.world
        dist.all_gather_object(parts, local, group=self.group)
        merged = {}
        for p in parts:
            merged.update(p)
        return merged
