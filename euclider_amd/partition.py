"""Row-tile partition of a frame over the GPUs of one node (host-side index arithmetic only).

A frame's rows are cut into 8-row strips and strip s goes to rank s % world (round-robin "row tiles":
glass objects cost hundreds of rays per pixel, so contiguous bands would be badly unbalanced).  Each rank
traces its strips into a compact buffer of local_rows(rank) rows; ONE gather collects the buffers on rank 0
and `gather_permutation` restores row order.  The same arithmetic lives in the C ABI (eu_frame.strip_count /
strip_index, eu_frame_local_rows) and in the gen kernel.
"""
STRIP = 8


def local_rows(height, rank, world):
    """Rows of rank's buffer (padded to whole strips), = eu_frame_local_rows for row range [0, height)."""
    if world <= 1:
        return height
    strips = (height + STRIP - 1) // STRIP
    mine = (strips + world - 1 - rank) // world
    return mine * STRIP


def global_row(local_row, rank, world):
    """Frame row traced into local_row of rank's buffer (may be >= height in the padded last strip)."""
    if world <= 1:
        return local_row
    return ((local_row // STRIP) * world + rank) * STRIP + local_row % STRIP


def gather_permutation(height, world, rows_per_rank):
    """perm[r] = row index inside the concatenated [rank0 | rank1 | ...] gather buffer (each rank padded to
    rows_per_rank rows) that holds frame row r."""
    perm = []
    for r in range(height):
        strip = r // STRIP
        perm.append((strip % world) * rows_per_rank + (strip // world) * STRIP + r % STRIP)
    return perm
