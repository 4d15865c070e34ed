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


def segment_by_length(seq, qual, offsets):
    """ragged -> the segmented C-ABI layout: reads grouped by length into tiles of <= 64 rows, each
    tile at its own stride (multiple of 8 with an odd number of 8-byte units).
    Returns (seq_bytes, qual_bytes, tiles[TILE_DTYPE], out_index, max_stride)."""
    from sickle_amd.capi import TILE_DTYPE
    lens = np.diff(offsets).astype(np.int64)
    order = np.argsort(lens, kind="stable")
    tiles = []
    qb, sb = [], []
    at = 0
    slot = 0
    i = 0
    n = len(order)
    max_stride = 8
    while i < n:
        L = int(lens[order[i]])
        j = i
        while j < n and lens[order[j]] == L:
            j += 1
        stride = ((L + 7) // 8 | 1) * 8
        max_stride = max(max_stride, stride)
        for a in range(i, j, 64):
            rows = min(64, j - a)
            pad = (-at) % 16
            if pad:
                qb.append(np.zeros(pad, dtype=np.uint8))
                sb.append(np.zeros(pad, dtype=np.uint8))
                at += pad
            qm = np.zeros((rows, stride), dtype=np.uint8)
            sm = np.zeros((rows, stride), dtype=np.uint8)
            for k in range(rows):
                o = int(offsets[order[a + k]])
                qm[k, :L] = qual[o:o + L]
                sm[k, :L] = seq[o:o + L]
            qb.append(qm.reshape(-1))
            sb.append(sm.reshape(-1))
            tiles.append((at, slot, stride, rows, L, 0))
            at += rows * stride
            slot += rows
        i = j
    return (np.concatenate(sb), np.concatenate(qb), np.array(tiles, dtype=TILE_DTYPE), order.astype(np.uint32), max_stride)
