"""Deterministic synthetic inputs shared by the golden generator and the tests (no reference code)."""
import struct

import numpy as np

MASK = (1 << 64) - 1


def splitmix64(n, seed):
    """n words of the splitmix64 stream started at `seed` (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_bloom_words(nwords, seed, mode):
    """Random bloom bit array with a chosen bit density: 'a|(b&c)' -> 0.625, 'a|b' -> 0.75, 'a' -> 0.5, 'a&(b|c)' -> 0.375."""
    a = splitmix64(nwords, seed * 3 + 1)
    b = splitmix64(nwords, seed * 3 + 2)
    c = splitmix64(nwords, seed * 3 + 3)
    if mode == "a|(b&c)":
        return a | (b & c)
    if mode == "a|b":
        return a | b
    if mode == "a":
        return a
    if mode == "a&(b|c)":  # 0.375: the bit density of a .blf at its design load (20 probes, 43 bits per entry)
        return a & (b | c)
    raise ValueError(mode)


def write_blf(path, words):
    """.blf container: u32 'ECBF' magic, u32 version 1, u64 size in words, then the words (utils.c:274-275,328-360)."""
    words = np.ascontiguousarray(words, dtype="<u8")
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQ", 0x45434246, 1, len(words)))
        f.write(words.tobytes())


def read_blf(path):
    raw = open(path, "rb").read()
    magic, ver, size = struct.unpack("<IIQ", raw[:16])
    if magic != 0x45434246 or ver != 1:
        raise ValueError("not a v1 .blf file")
    return np.frombuffer(raw, dtype="<u8", count=size, offset=16).copy()
