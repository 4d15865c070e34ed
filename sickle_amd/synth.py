"""Seeded synthetic FASTQ generator (the workloads of SURVEY.md section 8d).

Quality model per read: base Q ~ U[30,40], linear decay ~ U[0,0.25] per base, N(0, sigma=4)
noise, clipped to [2,41]; a low-quality 5' head of 0..7 bases (Q=2) on every read; a fraction
of bases replaced by 'N' (their quality forced to 2).  Encodings: sanger = phred+33,
illumina = phred+64 (clip floor raised so that chars stay in 64..110)."""
import numpy as np

OFFSET = {"sanger": 33, "illumina": 64, "solexa": 64}


def quality_matrix(rng, n, length, qualtype="sanger"):
    """(n, length) uint8 matrix of quality CHARACTERS."""
    base = rng.uniform(30.0, 40.0, size=(n, 1)).astype(np.float32)
    decay = rng.uniform(0.0, 0.25, size=(n, 1)).astype(np.float32)
    pos = np.arange(length, dtype=np.float32)[None, :]
    q = base - decay * pos + rng.normal(0.0, 4.0, size=(n, length)).astype(np.float32)
    q = np.clip(np.rint(q), 2, 41).astype(np.int16)
    head = rng.integers(0, 8, size=(n, 1))
    q[np.arange(length)[None, :] < head] = 2
    return (q + OFFSET[qualtype]).astype(np.uint8)


def sequence_matrix(rng, n, length, n_frac=0.002, lower_n_frac=0.0):
    """(n, length) uint8 matrix of bases; n_frac of them 'N'; lower_n_frac of READS get one 'n'."""
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, length))]
    seq = seq.copy()
    seq[rng.random(size=(n, length)) < n_frac] = ord("N")
    if lower_n_frac > 0:
        rows = np.nonzero(rng.random(n) < lower_n_frac)[0]
        seq[rows, rng.integers(0, length, size=rows.size)] = ord("n")
    return seq


def make_reads(seed, n, length, qualtype="sanger", n_frac=0.002, lower_n_frac=0.0):
    """Fixed-length reads: (seq[n,length], qual[n,length]) uint8."""
    rng = np.random.default_rng(seed)
    qual = quality_matrix(rng, n, length, qualtype)
    seq = sequence_matrix(rng, n, length, n_frac, lower_n_frac)
    qual[(seq == ord("N")) | (seq == ord("n"))] = 2 + OFFSET[qualtype]
    return seq, qual


def make_ragged_reads(seed, n, lo, hi, qualtype="illumina", n_frac=0.003, lower_n_frac=0.05):
    """Mixed-length reads lo..hi: (seq_bytes, qual_bytes, offsets[n+1] uint64), packed back to back."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=n)
    seq_m, qual_m = make_reads(seed + 1, n, hi, qualtype, n_frac, lower_n_frac)
    mask = np.arange(hi)[None, :] < lens[:, None]
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    return seq_m[mask], qual_m[mask], offsets


def make_long_reads(seed, n, lo, hi):
    """Long reads (Sanger chars), lengths log-uniform in lo..hi, of good quality with, read by read, a bad start,
    a collapse somewhere, a collapse in the last bases only, a stretch hovering at Q20, or none of these;
    an N or n in one read of three.  -> (seq_bytes, qual_bytes, offsets[n+1] uint64), packed back to back."""
    rng = np.random.default_rng(seed)
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n)).astype(np.int64)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    tot = int(offsets[-1])
    qual = np.clip(rng.normal(63, 5, tot).astype(int), 33, 74).astype(np.uint8)
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=tot)
    for i in range(n):
        a, b = int(offsets[i]), int(offsets[i + 1])
        mode = i % 5
        if mode == 0:
            qual[a:a + (b - a) // 4] = np.clip(rng.normal(41, 4, (b - a) // 4).astype(int), 33, 74)
        elif mode == 1:
            c = a + int(rng.integers(0, b - a))
            qual[c:b] = np.clip(rng.normal(41, 4, b - c).astype(int), 33, 74)
        elif mode == 2:
            qual[max(a, b - int(rng.integers(1, 60))):b] = 35
        elif mode == 3:
            c = a + int(rng.integers(0, b - a))
            qual[c:b] = 53 + rng.integers(-2, 3, size=b - c)
        if i % 3 == 0:
            seq[a + int(rng.integers(0, b - a))] = ord("N") if i % 2 else ord("n")
    return seq, qual, offsets


def pack_fixed(mat, stride):
    """Pad an (n, L) byte matrix to rows of `stride` bytes (zero fill) and flatten."""
    n, length = mat.shape
    out = np.zeros((n, stride), dtype=np.uint8)
    out[:, :length] = mat
    return out.reshape(-1)


def fastq_bytes(seq, qual, prefix="SYN:", start=0, suffix="", plus="+"):
    """FASTQ text for fixed-length matrices.  Header '@SYN:%09d' (+suffix), bare '+'."""
    n = seq.shape[0]
    parts = []
    for i in range(n):
        parts.append(b"@%s%09d%s\n" % (prefix.encode(), start + i, suffix.encode()))
        parts.append(seq[i].tobytes())
        parts.append(b"\n" + plus.encode() + b"\n")
        parts.append(qual[i].tobytes())
        parts.append(b"\n")
    return b"".join(parts)


def fastq_bytes_ragged(seq, qual, offsets, prefix="SYN:", start=0, suffix="", plus="+"):
    parts = []
    for i in range(len(offsets) - 1):
        a, b = int(offsets[i]), int(offsets[i + 1])
        parts.append(b"@%s%09d%s\n" % (prefix.encode(), start + i, suffix.encode()))
        parts.append(seq[a:b].tobytes())
        parts.append(b"\n" + plus.encode() + b"\n")
        parts.append(qual[a:b].tobytes())
        parts.append(b"\n")
    return b"".join(parts)


def fastq_bytes_fast(seq, qual, prefix="SYN:", start=0, suffix="", plus="+"):
    """Same bytes as fastq_bytes() for fixed-length matrices, built as one (n, record) byte matrix."""
    n, length = seq.shape
    head = ("@" + prefix).encode()
    tail = suffix.encode() + b"\n"
    mid = b"\n" + plus.encode() + b"\n"
    width = len(head) + 9 + len(tail) + length + len(mid) + length + 1
    rec = np.empty((n, width), dtype=np.uint8)
    c = 0
    rec[:, c:c + len(head)] = np.frombuffer(head, dtype=np.uint8)
    c += len(head)
    ids = np.arange(start, start + n, dtype=np.int64)
    for d in range(9):
        rec[:, c + 8 - d] = (ids % 10 + 48).astype(np.uint8)
        ids //= 10
    c += 9
    rec[:, c:c + len(tail)] = np.frombuffer(tail, dtype=np.uint8)
    c += len(tail)
    rec[:, c:c + length] = seq
    c += length
    rec[:, c:c + len(mid)] = np.frombuffer(mid, dtype=np.uint8)
    c += len(mid)
    rec[:, c:c + length] = qual
    c += length
    rec[:, c] = 10
    return rec.tobytes()
