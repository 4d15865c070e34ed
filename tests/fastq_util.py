"""Small FASTQ helpers for the tests (plain 4-line records)."""
import numpy as np


def parse_fastq(data):
    """bytes -> list of (name, seq, comment, qual) byte strings (lines without the newline)."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    assert len(lines) % 4 == 0, len(lines)
    return [tuple(lines[i:i + 4]) for i in range(0, len(lines), 4)]


def pack_records(records):
    """-> (seq_bytes, qual_bytes, offsets) packed back to back (the ragged C-ABI layout)."""
    lens = np.array([len(r[1]) for r in records], dtype=np.uint64)
    offsets = np.zeros(len(records) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    seq = np.frombuffer(b"".join(r[1] for r in records), dtype=np.uint8)
    qual = np.frombuffer(b"".join(r[3] for r in records), dtype=np.uint8)
    return seq, qual, offsets


def emit_records(records, cuts):
    """The record format of reference src/trim_single.cpp:393-396 for kept reads, in order."""
    out = []
    for (name, seq, comment, qual), (five, three) in zip(records, cuts):
        if three >= 0:
            out.append(name + b"\n" + seq[five:three] + b"\n" + comment + b"\n" + qual[five:three] + b"\n")
    return b"".join(out)
