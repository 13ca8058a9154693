"""Frame sharding across ranks: frames are independent units, rank r of `world`
takes a contiguous block; block starts are multiples of 4 so every shard keeps whole
frame-quads (the decoder's dword = 4 frames)."""


def shard_range(frames, rank, world):
    quads = (frames + 3) // 4
    lo = (quads * rank) // world * 4
    hi = (quads * (rank + 1)) // world * 4
    return min(lo, frames), min(hi, frames)
