"""Deterministic synthetic byte streams (integer-only, reproducible on any box).

These are the inputs named in SURVEY.md section 8(d): `random` (splitmix64), `text`
(Zipf-ish word model over [a-z] with space/newline separators) and `dna` (ACGT with
planted tandem repeats).  The splitmix64 stream is the one used for the hash
known-answers in SURVEY.md section 4.3, so `fnv1a64(random_bytes(4096, 1))` must equal
0xd09effa23070fc72.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15


def _splitmix64_block(seed: int, start: int, count: int) -> np.ndarray:
    """Draws number start+1 .. start+count of the splitmix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + k * np.uint64(_GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_bytes(n: int, seed: int = 12345, chunk: int = 1 << 24) -> np.ndarray:
    """n uniform random bytes: splitmix64 draws stored little-endian, last draw truncated."""
    out = np.empty(n, dtype=np.uint8)
    nd = (n + 7) // 8
    pos = 0
    for s in range(0, nd, chunk):
        c = min(chunk, nd - s)
        b = _splitmix64_block(seed, s, c).astype("<u8").view(np.uint8)
        take = min(b.size, n - pos)
        out[pos:pos + take] = b[:take]
        pos += take
    return out


def dna_bytes(n: int, seed: int = 7) -> np.ndarray:
    """Random ACGT, one base per byte (each random byte -> "ACGT"[b&3])."""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return lut[random_bytes(n, seed) & 3]


def dna_tandem_bytes(n: int, seed: int = 9) -> np.ndarray:
    """ACGT with planted tandem repeats: alternating random stretches (1..21k) and tandem
    blocks (unit 1..50, 10..2010 copies); integer-only."""
    out = np.empty(n, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    ctl = _splitmix64_block(seed ^ 0x5DEECE66D, 0, 4 * (n // 1000 + 16))
    ci = 0
    pos = 0
    blk = 0
    while pos < n:
        ln = 1 + int(ctl[ci] % np.uint64(21000)); ci += 1
        ln = min(ln, n - pos)
        out[pos:pos + ln] = lut[random_bytes(ln, seed + 1000 + blk) & 3]
        pos += ln
        blk += 1
        if pos >= n:
            break
        unit = 1 + int(ctl[ci] % np.uint64(50)); ci += 1
        copies = 10 + int(ctl[ci] % np.uint64(2001)); ci += 1
        u = lut[random_bytes(unit, seed + 500000 + blk) & 3]
        ln = min(unit * copies, n - pos)
        out[pos:pos + ln] = np.resize(u, ln)
        pos += ln
        if ci + 4 >= ctl.size:
            ci = 0
    return out


def text_bytes(n: int, seed: int = 3, vocab: int = 50000) -> np.ndarray:
    """English-like text: ~`vocab` words over [a-z] drawn with an integer Zipf-ish law
    (rank = floor(vocab * u^2 * u) with u in [0,1) fixed-point), separated by ' ' and
    occasionally '\\n'.  Integer-only so every box produces the same bytes."""
    # vocabulary: word lengths 1..12, letters from a skewed alphabet
    wl = (_splitmix64_block(seed ^ 0xABCDEF, 0, vocab) % np.uint64(12)).astype(np.int64) + 1
    offs = np.zeros(vocab + 1, dtype=np.int64)
    np.cumsum(wl, out=offs[1:])
    raw = random_bytes(int(offs[-1]), seed + 17)
    freq = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqzeeeettaaooiinn", dtype=np.uint8)
    letters = freq[raw % freq.size]
    out = np.empty(n + 16, dtype=np.uint8)
    pos = 0
    batch = 1 << 16
    it = 0
    while pos < n:
        r = _splitmix64_block(seed + 99, it * batch, batch)
        it += 1
        u = (r >> np.uint64(43)).astype(np.uint64)            # 21-bit fixed point
        rank = ((u * u >> np.uint64(21)) * u >> np.uint64(21)) * np.uint64(vocab) >> np.uint64(21)
        rank = rank.astype(np.int64)
        nl = ((r & np.uint64(15)) == 0)
        lens = wl[rank] + 1
        ends = np.cumsum(lens)
        total = int(ends[-1])
        buf = np.empty(total, dtype=np.uint8)
        starts = ends - lens
        # gather word letters
        idx = np.repeat(offs[rank] - starts, lens) + np.arange(total)
        sep_pos = ends - 1
        idx[sep_pos] = 0
        buf[:] = letters[np.minimum(idx, letters.size - 1)]
        buf[sep_pos] = np.where(nl, 10, 32).astype(np.uint8)
        take = min(total, n - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
    return out[:n].copy()


def sweep_bytes(alphabet: int, size: int) -> np.ndarray:
    """Deterministic stand-in for the demo self-test inputs rand()%alphabet
    (reference main.cpp:274-286, 389-435)."""
    return (random_bytes(size, seed=alphabet * 100003 + size) % alphabet).astype(np.uint8)


def fnv1a64(data) -> int:
    """Byte-wise FNV-1a-64 (offset 0xcbf29ce484222325, prime 0x100000001b3), pure python
    over a bytes-like; used for small known answers only (O(n) python loop)."""
    h = 0xCBF29CE484222325
    for b in memoryview(np.ascontiguousarray(data)).cast("B"):
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


GENERATORS = {
    "random": random_bytes,
    "dna": dna_bytes,
    "dna_tandem": dna_tandem_bytes,
    "text": text_bytes,
}
