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
    import bench                                               # the rank logic under test is bench.py's own
    from sickle_amd import synth
    from sickle_amd.shard import reduce_counters
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1"] + sys.argv[1:])
    plan = bench.job_plan(args, rank, world)
    n_job = args.total_reads if plan["strong"] else args.reads * world
    seq, qual = synth.make_reads(42, n_job, 150, "sanger")    # every rank can regenerate the job's reads
    b, e = plan["lo"], plan["lo"] + plan["n"]
    # the scan of this rank's shard (the oracle stands in for the GPU in this CPU test)
    cuts, err = ob.oracle_trim_batch(ob.make_params("sanger"), qual[b:e].reshape(-1), stride=150, read_len=150, n_reads=e - b)
    kept = int((cuts[:, 1] >= 0).sum())
    bases = int(np.clip(cuts[:, 1] - cuts[:, 0], 0, None).sum())
    counts, tmax = reduce_counters(dist, [kept, (e - b) - kept, bases, e - b], 0.5 + rank)
    if rank == 0:
        res = bench.headline(args, plan, world, counts, tmax, [1.0, 2.0, 3.0], "test")
        print(json.dumps({"counts": counts, "tmax": tmax, "line": res, "span": [b, e]}))
    dist.barrier()
    dist.destroy_process_group()
""")


def run_ranks(tmp_path, extra, port):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)] + extra,
                        capture_output=True, timeout=300, env=env)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    line = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def oracle_counts(n):
    import oracle_bind as ob
    from sickle_amd import synth
    seq, qual = synth.make_reads(42, n, 150, "sanger")
    cuts, _ = ob.oracle_trim_batch(ob.make_params("sanger"), qual.reshape(-1), stride=150, read_len=150, n_reads=n)
    kept = int((cuts[:, 1] >= 0).sum())
    return kept, int(np.clip(cuts[:, 1] - cuts[:, 0], 0, None).sum())


def test_two_ranks_strong_scaling_line(tmp_path):
    """bench.py --total-reads: the job's reads split into contiguous shards (BASELINE configs[3] shape), the
    counters of the whole job on rank 0's line, throughput = all reads x steps over the slowest rank's time."""
    got = run_ranks(tmp_path, ["--total-reads", "20001"], 29533)
    kept, bases = oracle_counts(20001)
    assert got["counts"] == [kept, 20001 - kept, bases, 20001]
    assert got["tmax"] == 1.5  # max over ranks
    line = got["line"]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["reads_in_job"] == 20001 and line["config"]["reads_per_gpu"] == 10001
    assert "sharded across 2" in line["config"]["workload"]
    assert abs(line["value"] - 20001 * 3 / 1.5) < 1e-6
    assert line["kept"] == kept and line["roofline"]["kernel_ms_avg"] == 2.0


def test_two_ranks_weak_scaling_line(tmp_path):
    got = run_ranks(tmp_path, ["--reads", "7001"], 29534)
    kept, _ = oracle_counts(14002)
    line = got["line"]
    assert got["span"] == [0, 7001]
    assert line["scaling"] == "weak" and line["config"]["reads_in_job"] == 14002 and line["kept"] == kept
    assert "configs[1]" in line["config"]["workload"]
