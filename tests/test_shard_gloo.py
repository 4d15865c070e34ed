"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- contiguous read shards per rank,
no data-path collective, one sum of counters / max of time at the end."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from sickle_amd.shard import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 64, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_bind as ob
    from sickle_amd import synth
    from sickle_amd.shard import shard_range, reduce_counters
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 20001
    seq, qual = synth.make_reads(42, n, 150, "sanger")        # every rank can regenerate the batch
    b, e = shard_range(n, rank, world)
    # the scan of this rank's shard (the oracle stands in for the GPU in this CPU test)
    cuts, err = ob.oracle_trim_batch(ob.make_params("sanger"), qual[b:e].reshape(-1), stride=150, read_len=150, n_reads=e - b)
    kept = int((cuts[:, 1] >= 0).sum())
    counts, tmax = reduce_counters(dist, [kept, (e - b) - kept], 0.5 + rank)
    if rank == 0:
        print(json.dumps({"counts": counts, "tmax": tmax}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_ranks_sum_counters(tmp_path):
    import json
    import oracle_bind as ob
    from sickle_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                        capture_output=True, timeout=300, env=env)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    line = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    seq, qual = synth.make_reads(42, 20001, 150, "sanger")
    cuts, _ = ob.oracle_trim_batch(ob.make_params("sanger"), qual.reshape(-1), stride=150, read_len=150, n_reads=20001)
    kept = int((cuts[:, 1] >= 0).sum())
    assert got["counts"] == [kept, 20001 - kept]
    assert got["tmax"] == 1.5  # max over ranks
