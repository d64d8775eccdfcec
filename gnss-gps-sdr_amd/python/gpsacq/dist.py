"""Multi-GPU sharding of the search grid (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference is single-threaded (c/search_offline.cpp has no parallelism at all); every
(block, PRN, Doppler) cell is independent, so the grid shards without any data-path
collective.  Two decompositions are provided:

  * by block  (reference schedule, SearchTask :239-246: block b <-> PRN b % 32): rank r searches
    a contiguous range of whole runs; nothing is exchanged but the results.
  * by Doppler slab / PRN (one block against the whole PRN x Doppler grid): each rank searches a
    subset of the (PRN, Doppler-range) pairs and the per-PRN best peak is combined with ONE
    all-reduce(MAX) of 32 packed 64-bit keys (256 bytes) -- the only exchange step of the path.

Key packing: for non-negative IEEE floats the bit pattern orders like the value, so
    key = snr_bits << 32 | (0xFFFF - doppler_index) << 16 | ca_shift
makes integer MAX pick the higher SNR and, on equal SNR, the LOWER Doppler bin -- the
reference's strict '>' first-wins rule while scanning dop upwards (:196-198).
"""
import torch

NUM_SATS = 32


def shard_runs(n_runs, rank, world):
    """Contiguous, balanced split of whole runs (32 blocks each): returns (first_run, n)."""
    base, rem = divmod(n_runs, world)
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def shard_doppler(dmax, rank, world):
    """Contiguous split of the Doppler bins -dmax..+dmax: returns (first_bin, n_bins)."""
    first, n = shard_runs(2 * dmax + 1, rank, world)
    return first - dmax, n


def shard_doppler_grid(n_points, first_point, rank, world):
    """Contiguous split of a Doppler grid of n_points points starting at index first_point (Engine.num_doppler_total,
    .first_doppler_total): returns (first, n); n may be 0 when there are more ranks than points."""
    first, n = shard_runs(n_points, rank, world)
    return first + first_point, n


def pack_keys(peaks_i32, dmax):
    """peaks_i32: int32 tensor [n, 4] viewing gpsacq_peak records {snr f32 bits, lo, ca, max_pwr bits}."""
    snr_bits = peaks_i32[:, 0].to(torch.int64) & 0xFFFFFFFF
    lo = peaks_i32[:, 1].to(torch.int64) + dmax
    ca = peaks_i32[:, 2].to(torch.int64) & 0xFFFF
    return (snr_bits << 32) | ((0xFFFF - lo) << 16) | ca


def unpack_keys(keys, dmax):
    snr = ((keys >> 32) & 0xFFFFFFFF).to(torch.int32).view(torch.float32) if keys.dtype == torch.int64 else None
    lo = 0xFFFF - ((keys >> 16) & 0xFFFF) - dmax
    ca = keys & 0xFFFF
    return snr, lo.to(torch.int32), ca.to(torch.int32)


def per_prn_best(keys, prn_of_task=None):
    """Best key per PRN.  Reference schedule (prn_of_task None): task t <-> PRN t % 32."""
    if prn_of_task is None:
        return keys.view(-1, NUM_SATS).max(dim=0).values
    best = torch.zeros(NUM_SATS, dtype=torch.int64, device=keys.device)
    return best.scatter_reduce(0, prn_of_task.to(torch.int64), keys, reduce="amax", include_self=True)


def allreduce_best(best, group=None):
    """The path's one collective: all-reduce(MAX) of the 32 per-PRN keys."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(best, op=dist.ReduceOp.MAX, group=group)
    return best


def gather_peaks(peaks_i32, group=None):
    """All ranks' peak records in rank order (for the full SearchTask report); ragged-safe."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return peaks_i32
    world = dist.get_world_size(group)
    n = torch.tensor([peaks_i32.shape[0]], dtype=torch.int64, device=peaks_i32.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx, 4), dtype=peaks_i32.dtype, device=peaks_i32.device)
    pad[:peaks_i32.shape[0]] = peaks_i32
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:int(s.item())] for o, s in zip(out, sizes)], dim=0)
