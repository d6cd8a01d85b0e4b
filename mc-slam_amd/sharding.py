"""Camera / frame placement for the multi-GPU path, pure index arithmetic.

One process per GPU.  A step handles `frames_per_rank * world` rig frames:
  * extraction of camera c of frame f runs on rank (c + f) mod world  (one camera per GPU
    when world == ncams; rotating with f so every rank carries the same load),
  * each rank exports its descriptor sets in (frame, camera) order; ONE all-gather makes
    every rank hold all sets: set index = owner * sets_per_rank + local position,
  * frame f is matched (all camera pairs + track merge) on rank f mod world.

The all-gather lands every rank's sets on every rank although a rank only matches 1/world of the frames.  The
`a2a_*` functions describe the exchange that moves exactly what the matching needs (an all-to-all with uneven
splits): a rank orders its images by (destination rank, frame, camera), so its export buffer is already grouped
by destination, and receives, from every source rank in turn, the sets of the frames it matches.
"""
import numpy as np


def owner(c, f, world):
    return (c + f) % world


def images_of_rank(rank, world, ncams, total_frames):
    """(frame, camera) pairs rank `rank` extracts, in its local image order."""
    return [(f, c) for f in range(total_frames) for c in range(ncams) if owner(c, f, world) == rank]


def sets_per_rank(world, ncams, total_frames):
    n = [len(images_of_rank(r, world, ncams, total_frames)) for r in range(world)]
    if len(set(n)) != 1:
        raise ValueError("unbalanced placement: total_frames*ncams must split evenly over %d ranks (got %s)" % (world, n))
    return n[0]


def gathered_set_index(world, ncams, total_frames):
    """dict (frame, camera) -> set index inside the all-gathered descriptor block."""
    per = sets_per_rank(world, ncams, total_frames)
    idx = {}
    for r in range(world):
        for i, fc in enumerate(images_of_rank(r, world, ncams, total_frames)):
            idx[fc] = r * per + i
    return idx


def frames_of_rank(rank, world, total_frames):
    return [f for f in range(total_frames) if f % world == rank]


def match_sets(rank, world, ncams, total_frames):
    """(frames matched on this rank, int32 array [nframes][ncams] of gathered set indices)."""
    idx = gathered_set_index(world, ncams, total_frames)
    fr = frames_of_rank(rank, world, total_frames)
    return fr, np.array([[idx[(f, c)] for c in range(ncams)] for f in fr], np.int32).reshape(len(fr), ncams)


# ---- all-to-all: every set travels only to the rank that matches its frame ----
def dest(f, world):
    return f % world


def a2a_images_of_rank(rank, world, ncams, total_frames):
    """(frame, camera) pairs rank `rank` extracts, ordered by (destination rank, frame, camera): the local image order,
    which makes the exported descriptor block a ready-made all-to-all send buffer."""
    return sorted(images_of_rank(rank, world, ncams, total_frames), key=lambda fc: (dest(fc[0], world), fc[0], fc[1]))


def a2a_send_splits(rank, world, ncams, total_frames):
    """number of sets rank `rank` sends to each destination rank"""
    n = [0] * world
    for f, _ in images_of_rank(rank, world, ncams, total_frames):
        n[dest(f, world)] += 1
    return n


def a2a_recv_splits(rank, world, ncams, total_frames):
    """number of sets rank `rank` receives from each source rank"""
    return [a2a_send_splits(src, world, ncams, total_frames)[rank] for src in range(world)]


def a2a_match_sets(rank, world, ncams, total_frames):
    """(frames matched on this rank, int32 [nframes][ncams] of set indices inside the RECEIVED block): the block is the
    concatenation over source ranks of what each sends here, every part in the sender's (frame, camera) order."""
    idx, base = {}, 0
    for src in range(world):
        part = [fc for fc in a2a_images_of_rank(src, world, ncams, total_frames) if dest(fc[0], world) == rank]
        for i, fc in enumerate(part):
            idx[fc] = base + i
        base += len(part)
    fr = frames_of_rank(rank, world, total_frames)
    return fr, np.array([[idx[(f, c)] for c in range(ncams)] for f in fr], np.int32).reshape(len(fr), ncams)


# ---- all-gather + pair partition: the split SURVEY.md 8(e) describes for one rig frame across GPUs ----
# Every rank holds every set after the all-gather (gathered_set_index); BruteForceMatch of camera pair (i, j) of frame f runs on
# rank (i + j + f) mod world -- 8(e)'s (i + j) mod G, rotated with the frame so that the six pairs of a 4-camera frame, which
# never divide evenly, even out over consecutive frames -- and the accepted (query, train) lists go to rank 0, which runs
# computeIntraMatches' serial merge (MultiCameraFrame.cpp:1167-1268).
def pair_owner(i, j, f, world):
    return (i + j + f) % world


def pairs_of_rank(rank, world, ncams, total_frames):
    """(frame, i, j), i < j, matched on `rank`, in (frame, i, j) order"""
    return [(f, i, j) for f in range(total_frames) for i in range(ncams - 1) for j in range(i + 1, ncams) if pair_owner(i, j, f, world) == rank]


def pair_batches(rank, world, ncams, total_frames, frames_per_batch):
    """the rank's pairs cut into jobs of `frames_per_batch` consecutive frames (a job may name at most one slot's worth of
    distinct sets): list of (pairs [(f, i, j)], int32 [npairs][2] of gathered set indices)"""
    idx = gathered_set_index(world, ncams, total_frames)
    mine = pairs_of_rank(rank, world, ncams, total_frames)
    out = []
    for f0 in range(0, total_frames, frames_per_batch):
        part = [p for p in mine if f0 <= p[0] < f0 + frames_per_batch]
        if part:
            out.append((part, np.array([[idx[(f, i)], idx[(f, j)]] for f, i, j in part], np.int32).reshape(len(part), 2)))
    return out


def pair_slot(ncams, i, j):
    """index of camera pair (i, j), i < j, in (0,1), (0,2), .., (1,2), .. order"""
    return i * ncams - i * (i + 1) // 2 + (j - i - 1)
