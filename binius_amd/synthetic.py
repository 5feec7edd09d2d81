"""Synthetic inputs for benchmarks and tools: the documented SplitMix64 streams
in numpy.  (The oracle has the same generator in C for the tests; the product side does not import
the oracle.)"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix_words(seed, n_words):
    """n_words 64-bit outputs of SplitMix64 seeded with `seed` (state += gamma before every draw)."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + _GAMMA * np.arange(1, n_words + 1, dtype=np.uint64)
        z = (x ^ (x >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def random_b128(seed, n):
    """n uniform BinaryField128b elements as an (n, 2) uint64 array (lo, hi): two draws per element."""
    return splitmix_words(seed, 2 * n).reshape(n, 2)


def random_b128_shard(seed, n_local, world, rank, start=0):
    """Rank `rank`'s shard of the GLOBAL array random_b128(seed, n_local * world): the elements with global
    index = rank (mod world), in local order (binius_amd/distributed.py shard_indices), computed without
    materialising the global array -- SplitMix64 output k depends only on seed + (k + 1) * gamma.
    start: first local index (for chunked uploads)."""
    with np.errstate(over="ignore"):
        i = (np.uint64(start) + np.arange(n_local, dtype=np.uint64)) * np.uint64(world) + np.uint64(rank)  # global element index
        k = np.empty(2 * n_local, dtype=np.uint64)
        k[0::2] = np.uint64(2) * i + np.uint64(1)
        k[1::2] = np.uint64(2) * i + np.uint64(2)
        x = np.uint64(seed) + _GAMMA * k
        z = (x ^ (x >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return (z ^ (z >> np.uint64(31))).reshape(n_local, 2)


def random_scalars(seed, n):
    a = random_b128(seed, n)
    return [int(a[i, 0]) | (int(a[i, 1]) << 64) for i in range(n)]
