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


# ---- the reference's batching and -a T output order, restated for the tests --------------------------
def reference_batch_len(file_size, batch_mib=512, paired=False):
    """recommended_batch_len: reference src/trim_single.cpp:194-210, src/trim_paired.cpp:246-262"""
    mx = (batch_mib * 1024 * 1024) // (2 if paired else 1)
    return max(20, min(file_size // 8, mx))


def reference_batches(lines, batch_len, min_lines=4):
    """GZReader::read_lines + get_batch_buffering_lines (reference src/GZReader.cpp:29-41,59-132) on a file that
    ends with a newline: lines are taken until their lengths (without newline) have used up batch_len, lines
    beyond a multiple of min_lines are carried into the next batch, a batch without a complete record ends the
    input, lines still carried at EOF are dropped.  -> list of line lists."""
    out, carried, pos, eof = [], [], 0, False
    while not eof:
        cur = list(carried)
        remaining = batch_len - sum(len(l) for l in carried)
        carried = []
        while True:
            if pos >= len(lines):
                eof = True
                break
            cur.append(lines[pos])
            remaining -= len(lines[pos])
            pos += 1
            if remaining <= 0:
                break
        extra = len(cur) % min_lines
        if extra and cur:
            carried = cur[-extra:]
            cur = cur[:-extra]
        if not cur:
            break
        out.append(cur)
    return out


def file_lines(data):
    lines = data.split(b"\n")
    assert lines[-1] == b"", "test inputs end with a newline"
    return lines[:-1]


def record_text(rec, cut):
    """reference src/trim_paired.cpp:506-513"""
    name, seq, comment, qual = rec
    five, three = int(cut[0]), int(cut[1])
    return name + b"\n" + seq[five:three] + b"\n" + comment + b"\n" + qual[five:three] + b"\n"


def expected_pe_outputs(batches1, batches2, cut_of, threads, interleaved=False):
    """What `sickle pe -a T` writes when its batches come out in input order: per batch, pair k goes to queue
    k mod T (reference src/trim_paired.cpp:388-403) and the queues are written one after the other (:530-567).
    batches1/batches2: line lists per batch (batches2 None for interleaved input); cut_of(file, record) -> (five,
    three) with record = that file's running record number.  -> per-batch chunks [(out1, out2, singles)]."""
    chunks, r1, r2 = [], 0, 0
    for b, lines1 in enumerate(batches1):
        if interleaved:
            recs = [tuple(lines1[i:i + 4]) for i in range(0, len(lines1), 4)]
            pairs = [(recs[2 * k], recs[2 * k + 1], (0, r1 + 2 * k), (0, r1 + 2 * k + 1)) for k in range(len(recs) // 2)]
            r1 += len(recs)
        else:
            lines2 = batches2[b]
            assert len(lines2) == len(lines1)
            ra = [tuple(lines1[i:i + 4]) for i in range(0, len(lines1), 4)]
            rb = [tuple(lines2[i:i + 4]) for i in range(0, len(lines2), 4)]
            pairs = [(ra[k], rb[k], (0, r1 + k), (1, r2 + k)) for k in range(len(ra))]
            r1 += len(ra)
            r2 += len(rb)
        o1, o2, os_ = [], [], []
        for q in range(threads):
            for k in range(q, len(pairs), threads):
                a, b_, ia, ib = pairs[k]
                ca, cb = cut_of(*ia), cut_of(*ib)
                ka, kb = ca[1] >= 0, cb[1] >= 0
                if ka and kb:
                    o1.append(record_text(a, ca))
                    (o1 if interleaved else o2).append(record_text(b_, cb))
                elif ka:
                    os_.append(record_text(a, ca))
                elif kb:
                    os_.append(record_text(b_, cb))
        chunks.append((b"".join(o1), b"".join(o2), b"".join(os_)))
    return chunks


def expected_se_output(batches, cut_of, threads):
    """`sickle se -a T`: read k of a batch goes to queue (k + 1) mod T (reference src/trim_single.cpp:263-298),
    queues written in turn (:382-405).  -> per-batch chunks."""
    chunks, r0 = [], 0
    for lines in batches:
        recs = [tuple(lines[i:i + 4]) for i in range(0, len(lines), 4)]
        out = []
        for q in range(threads):
            for k in range((q + threads - 1) % threads, len(recs), threads):
                c = cut_of(0, r0 + k)
                if c[1] >= 0:
                    out.append(record_text(recs[k], c))
        r0 += len(recs)
        chunks.append(b"".join(out))
    return chunks


def is_permutation_of_chunks(data, chunks):
    """Is `data` the concatenation of all `chunks` in SOME order?  (The reference's per-batch output threads race
    for the files, so its batches land in any order; inside a batch the order is fixed.)"""
    left = [c for c in chunks if c]
    pos = 0
    while left:
        for i, c in enumerate(left):
            if data.startswith(c, pos):
                pos += len(c)
                del left[i]
                break
        else:
            return False
    return pos == len(data)
