"""Counter-based synthetic fields and order-independent checksums shared by the golden
generator, the CPU tests and the GPU tests.  The same integer recipe is implemented on
the device by mg_fill_uniform / mg_checksum (include/mg_hip.h), so full-size inputs
never have to cross PCIe and full-size outputs are compared through 128 bits of
checksum instead of gigabytes of fixtures."""
import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_G = np.uint64(0x9E3779B97F4A7C15)
_A = np.uint64(0xBF58476D1CE4E5B9)
_B = np.uint64(0x94D049BB133111EB)


def hash_uniform(start, count, seed):
    """uniform [0,1) doubles for flat indices start..start+count-1 (splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        z = (np.arange(start, start + count, dtype=np.uint64) + np.uint64(seed)) * _G
        z ^= z >> np.uint64(30)
        z *= _A
        z ^= z >> np.uint64(27)
        z *= _B
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_field(N, seed, chunk_rows=1024):
    out = np.empty((N, N), dtype=np.float64)
    flat = out.reshape(-1)
    for r0 in range(0, N, chunk_rows):
        r1 = min(N, r0 + chunk_rows)
        flat[r0 * N:r1 * N] = hash_uniform(r0 * N, (r1 - r0) * N, seed)
    return out


def checksum(a, chunk=1 << 24):
    """(sum of bit patterns, sum of bit patterns * (2*index+1)), both mod 2^64.
    -0.0 is canonicalised to +0.0 first (the driver's sign flip of D makes the sign of
    exact zeros an artefact)."""
    flat = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    s0 = np.uint64(0)
    s1 = np.uint64(0)
    with np.errstate(over="ignore"):
        for i0 in range(0, flat.size, chunk):
            part = flat[i0:i0 + chunk] + 0.0  # -0.0 + 0.0 == +0.0
            bits = part.view(np.uint64)
            idx = np.arange(i0, i0 + bits.size, dtype=np.uint64)
            s0 = s0 + bits.sum(dtype=np.uint64)
            s1 = s1 + (bits * (idx * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64)
    return int(s0), int(s1)
