"""GzInflater (sickle_amd/csrc/host/GzInflater.cpp), the gzip decoder of the ingest, against zlib
on the same bytes: every compression level and strategy, stored / fixed / dynamic blocks, long
Huffman codes (subtables), matches at the far end of the window, concatenated members with trailing
garbage, all optional header fields, odd read sizes, and randomly damaged or truncated files (no
crash, never silently wrong bytes).  CPU only; tests/cpu_shim/inflate_check is test infrastructure."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import cli_util as cu

BIN = os.path.join(cu.ROOT, "tests", "cpu_shim", "inflate_check")
PAR = os.path.join(cu.ROOT, "tests", "cpu_shim", "parallel_check")
BGZF = os.path.join(cu.ROOT, "tests", "cpu_shim", "bgzf_check")
TEXT = open(os.path.join(cu.INPUTS, "test.fastq"), "rb").read()


@pytest.fixture(scope="module")
def inflate_check():
    subprocess.run(["make", "-s", "-C", os.path.join(cu.ROOT, "tests", "cpu_shim"), "all"], check=True)
    return BIN


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8, wbits=31):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def with_all_header_fields(data):
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = body.compress(data) + body.flush()
    h = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\x03" + struct.pack("<H", 5) + b"ab\x01\x00z" + b"name.fq\0" + b"a comment\0"
    h += struct.pack("<H", zlib.crc32(h) & 0xffff)
    return h + raw + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def cases():
    rng = np.random.default_rng(5)
    rnd = rng.integers(0, 256, 1_000_000, dtype=np.uint8).tobytes()
    skew = rng.choice(np.arange(256, dtype=np.uint8), 1_000_000, p=np.r_[[0.5, 0.2, 0.1], np.full(253, 0.2 / 253)]).tobytes()
    zipf = (rng.zipf(1.3, 1_000_000) % 256).astype(np.uint8).tobytes()
    far = rng.integers(0, 256, 32700, dtype=np.uint8).tobytes()
    out = {"empty": (gz(b""), b""), "one": (gz(b"A"), b"A")}
    for lv in range(10):
        out["level%d" % lv] = (gz(TEXT, lv), TEXT)
    for name, st in (("fixed", zlib.Z_FIXED), ("huffman_only", zlib.Z_HUFFMAN_ONLY), ("rle", zlib.Z_RLE), ("filtered", zlib.Z_FILTERED)):
        out[name] = (gz(TEXT, 6, st), TEXT)
    out["memlevel1"] = (gz(TEXT * 2, 9, memlevel=1), TEXT * 2)
    out["window512"] = (gz(TEXT, 9, wbits=16 + 9), TEXT)
    out["random"] = (gz(rnd), rnd)
    out["random_stored"] = (gz(rnd, 0), rnd)
    runs = b"A" * 1_000_000 + b"BC" * 500_000 + b"xyz" * 300_000 + b"\0" * 70000
    out["runs"] = (gz(runs), runs)
    out["long_codes"] = (gz(skew, 9), skew)
    out["zipf"] = (gz(zipf, 9), zipf)
    out["far_matches"] = (gz(far * 30, 9), far * 30)
    out["members"] = (gz(TEXT) + gz(b"") + gz(rnd[:100000], 0) + gz(TEXT[:12345], 9) + b"\0\0\0trailing garbage",
                      TEXT + rnd[:100000] + TEXT[:12345])
    out["header_fields"] = (with_all_header_fields(TEXT[:50000]) + gz(TEXT[50000:90000]), TEXT[:90000])
    out["many_chunks"] = (gz(TEXT * 12), TEXT * 12)
    # hundreds of small members (what a BGZF file looks like to a reader that ignores its size
    # fields): a stretch has to step over member trailers and headers, and every block is a final one
    zeros = b"\0" * 60_000_000  # 1000:1 -- the parallel decoder must not hold it all as symbols at once
    out["very_compressible"] = (gz(zeros, 9), zeros)
    out["small_members"] = (b"".join(gz(TEXT[i * 20000:(i + 1) * 20000], 6) for i in range(41)) * 8, TEXT[:820000] * 8)
    return out


def test_decodes_like_zlib(inflate_check, tmp_path):
    for name, (blob, want) in cases().items():
        assert name == "members" or gzip.decompress(blob) == want  # (python's gzip rejects trailing garbage; zlib's gzread and the reference do not)
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(blob)
        for piece in (32 << 20, 1, 7, 4096, 65537, 300000):
            if piece < 100 and len(want) > 200000:
                continue
            pr = subprocess.run([inflate_check, path, str(piece)], capture_output=True)
            assert pr.returncode == 0 and pr.stdout == want, (name, piece, pr.stderr[:200])


def test_damaged_input_is_reported(inflate_check, tmp_path):
    rng = np.random.default_rng(9)
    blob = gz(TEXT)
    path = str(tmp_path / "dmg.gz")
    silent_ok = 0
    for k in range(150):
        b = bytearray(blob)
        at = int(rng.integers(10, len(b)))
        b[at] ^= 1 << int(rng.integers(0, 8))
        if k % 3 == 0:
            b = b[:at]
        open(path, "wb").write(bytes(b))
        pr = subprocess.run([inflate_check, path], capture_output=True)
        assert pr.returncode in (0, 2), (k, at, pr.returncode)  # 2 = error reported; anything else is a crash
        if pr.returncode == 0:
            assert pr.stdout == TEXT, (k, at)  # e.g. a flipped bit in the ISIZE-irrelevant header bytes
            silent_ok += 1
        else:
            assert b"error:" in pr.stderr
    assert silent_ok < 10


def test_parallel_decoder_like_zlib(inflate_check, tmp_path):
    """GzParallel on the same files, with chunks far smaller than in production so that every file
    takes many rounds of several stretches: block starts guessed inside stored, fixed and dynamic
    data, matches reaching into the unknown window, members ending mid-round."""
    used = dropped = rounds = 0
    for name, (blob, want) in cases().items():
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(blob)
        for chunk, width, piece in ((4096, 8, 1 << 20), (30000, 3, 65537), (1 << 20, 4, 1 << 25), (70000, 1, 1 << 22)):
            pr = subprocess.run([PAR, path, str(chunk), str(width), str(piece)], capture_output=True)
            assert pr.returncode == 0 and pr.stdout == want, (name, chunk, width, pr.stderr[-200:])
            f = pr.stderr.split()
            rounds += int(f[1])
            used += int(f[3])
            dropped += int(f[5])
            if name == "small_members" and chunk >= 30000:
                assert int(f[1]) < 328 // 4, ("rounds", f[1])  # far fewer rounds than members
    print("rounds", rounds, "stretches used", used, "dropped", dropped)
    assert used > 1.5 * rounds  # (width-1 runs and chunks smaller than a block count one per round) the stretches really were joined, not decoded serially
    assert dropped < used // 10


def test_parallel_decoder_damaged_input(inflate_check, tmp_path):
    rng = np.random.default_rng(19)
    blob = gz(TEXT * 3)
    path = str(tmp_path / "dmgp.gz")
    clean = 0
    for k in range(120):
        b = bytearray(blob)
        at = int(rng.integers(10, len(b)))
        b[at] ^= 1 << int(rng.integers(0, 8))
        if k % 3 == 0:
            b = b[:at]
        open(path, "wb").write(bytes(b))
        pr = subprocess.run([PAR, path, "20000", "6"], capture_output=True)
        assert pr.returncode in (0, 2), (k, at, pr.returncode)
        if pr.returncode == 0:
            assert pr.stdout == TEXT * 3, (k, at)
            clean += 1
        else:
            assert b"error:" in pr.stderr
    assert clean < 10


def test_fast_encoder_round_trips(inflate_check, tmp_path):
    """FqDeflate (SICKLE_GZ_LEVEL=fast) on FASTQ and on everything else a block may hold: each file
    -> BGZF -> zlib and both of this repo's decoders must give the bytes back.  Degenerate blocks:
    empty, one byte, one symbol only, incompressible (stored fallback), no newline at all, very
    long lines (the line four up is out of the window), counts skewed enough to need the 15-bit
    code length limit."""
    rng = np.random.default_rng(23)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    skew = b"".join(bytes([65 + i]) * f for i, f in enumerate(fib))  # Fibonacci counts: deepest possible tree
    files = {
        "fastq": TEXT,
        "fastq_crlf": TEXT[:200000].replace(b"\n", b"\r\n"),
        "empty": b"",
        "one": b"x",
        "same": b"a" * 200000,
        "random": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
        "noline": rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 200000).tobytes(),
        "longlines": b"\n".join(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 40000).tobytes() for _ in range(8)),
        "fibonacci": bytes(rng.permutation(np.frombuffer(skew, dtype=np.uint8))),
        "headers_only": b"".join(b"@M0:%d:FC:1:%d:%d:%d 1:N:0:ACGT\n" % (i % 7, 1100 + i // 999, 9 * i % 30000, 13 * i % 2500) for i in range(9000)),
    }
    for name, data in files.items():
        src = str(tmp_path / (name + ".txt"))
        open(src, "wb").write(data)
        for level in ("-1", "1"):
            pr = subprocess.run([BGZF, src, level], capture_output=True)
            assert pr.returncode == 0, (name, pr.stderr)
            blob = pr.stdout
            assert gzip.decompress(blob) == data, (name, level)
            gzp = str(tmp_path / (name + ".gz"))
            open(gzp, "wb").write(blob)
            for tool, args in ((inflate_check, []), (PAR, ["30000", "4"])):
                back = subprocess.run([tool, gzp] + args, capture_output=True)
                assert back.returncode == 0 and back.stdout == data, (name, level, tool, back.stderr[-200:])
        if name in ("fastq", "headers_only", "same"):
            assert len(blob) < 0.6 * len(data)


SIM = os.path.join(cu.ROOT, "tests", "cpu_shim", "gpu_deflate_sim")


def _encoder_inputs():
    rng = np.random.default_rng(23)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    skew = b"".join(bytes([65 + i]) * f for i, f in enumerate(fib))
    return {
        "fastq": TEXT,
        "fastq_crlf": TEXT[:200000].replace(b"\n", b"\r\n"),
        "one": b"x",
        "newline": b"\n",
        "same": b"a" * 200000,
        "random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "noline": rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 200000).tobytes(),
        "longlines": b"\n".join(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), 40000).tobytes() for _ in range(8)),
        "fibonacci": bytes(rng.permutation(np.frombuffer(skew, dtype=np.uint8))),
        "manylines": b"\n" * 70000 + b"a\n" * 40000 + b"\n\n\nxyz",
        "exact_block": TEXT[:65280],
        "block_plus_one": TEXT[:65281],
        "runs": b"A" * 258 + b"\n" + b"B" * 259 + b"\n" + b"C" * 600 + b"\n" + b"D" * 5 + b"E" * 4 + b"\n",
    }


def test_gpu_block_encoder_on_the_host(inflate_check, tmp_path):
    """sk_deflate_block.h, the phases the GPU's BGZF encoder runs per block, executed on the host
    lane after lane: zlib and both decoders here must give the text back."""
    for name, data in _encoder_inputs().items():
        src = str(tmp_path / (name + ".txt"))
        open(src, "wb").write(data)
        pr = subprocess.run([SIM, src], capture_output=True)
        assert pr.returncode == 0, (name, pr.stderr)
        assert gzip.decompress(pr.stdout) == data, name
        gzp = str(tmp_path / (name + ".gz"))
        open(gzp, "wb").write(pr.stdout)
        back = subprocess.run([inflate_check, gzp], capture_output=True)
        assert back.returncode == 0 and back.stdout == data, (name, back.stderr[-200:])


@pytest.mark.gpu
def test_gpu_block_encoder_matches_host_run(inflate_check, tmp_path):
    """sk_bgzf_deflate on the device: byte for byte what the host run of the same phases writes (the
    arithmetic is integer and the lanes' bits meet through atomic OR: order-independent), and zlib
    inflates it to the text."""
    from sickle_amd import capi
    for name, data in _encoder_inputs().items():
        src = str(tmp_path / (name + ".txt"))
        open(src, "wb").write(data)
        want = subprocess.run([SIM, src], capture_output=True).stdout
        got = capi.bgzf_deflate(data)
        assert gzip.decompress(got) == data, name
        assert got == want, name


@pytest.mark.gpu
def test_gpu_block_encoder_arguments_and_many_blocks():
    """sk_bgzf_deflate: nothing to do is not an error, bad arguments are, and a batch of a few
    thousand blocks (more than one round of the persistent grid: 5 workgroups x 256 CUs) inflates
    to the text."""
    from sickle_amd import capi
    L = capi.lib()
    assert L.sk_bgzf_deflate(0, None, None, 0, None, None) == 0
    assert L.sk_bgzf_deflate(0, None, None, 3, None, None) != 0
    assert L.sk_bgzf_deflate(99, None, None, 0, None, None) != 0
    data = TEXT * 130  # ~1640 blocks
    blob = capi.bgzf_deflate(data)
    assert gzip.decompress(blob) == data and len(blob) < 0.5 * len(data)
