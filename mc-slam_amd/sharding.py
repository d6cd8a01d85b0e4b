"""Camera / frame placement for the multi-GPU path, pure index arithmetic.

One process per GPU.  A step handles `frames_per_rank * world` rig frames:
  * extraction of camera c of frame f runs on rank (c + f) mod world  (one camera per GPU
    when world == ncams; rotating with f so every rank carries the same load),
  * each rank exports its descriptor sets in (frame, camera) order; ONE all-gather makes
    every rank hold all sets: set index = owner * sets_per_rank + local position,
  * frame f is matched (all camera pairs + track merge) on rank f mod world.
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
