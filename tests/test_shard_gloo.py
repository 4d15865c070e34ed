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
    from bench import wl
    from sickle_amd.shard import reduce_counters
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1"] + sys.argv[1:])
    plan = bench.job_plan(args, rank, world)
    cpu = torch.device("cpu")
    # this rank's reads, from bench.py's own generators (seeded per global read block), scanned by the oracle
    # (which stands in for the GPU in this CPU test)
    if args.workload == "mixed":
        lens, qual, seq = wl.mixed_shard(torch, cpu, bench.SEED + 4, plan["lo"], plan["n"])
        cuts, err = ob.oracle_trim_batch(ob.make_params("illumina", 20, 20, False, True), qual.numpy().reshape(-1), seq.numpy().reshape(-1),
                                         stride=wl.MIX_HI, lengths=lens.numpy().astype(np.uint32), n_reads=plan["n"])
        bases_in, algo = int(lens.sum()), 2 * int(lens.sum()) + 8 * plan["n"]
    else:
        qual = wl.se_shard(torch, cpu, bench.SEED, plan["lo"], plan["n"], 150, plan["stride"])
        cuts, err = ob.oracle_trim_batch(ob.make_params("sanger"), qual.numpy().reshape(-1), stride=plan["stride"], read_len=150, n_reads=plan["n"])
        bases_in, algo = 150 * plan["n"], 158 * plan["n"]
    assert err is None
    kept = int((cuts[:, 1] >= 0).sum())
    bases = int(np.clip(cuts[:, 1] - cuts[:, 0], 0, None).sum())
    counts, tmax = reduce_counters(dist, [kept, plan["n"] - kept, bases, plan["n"], bases_in], 0.5 + rank)
    h2d = bench.gather_floats(torch, dist, 10.0 + rank, None) if args.mode == "pipeline" else None
    if rank == 0:
        res = bench.headline(args, plan, world, counts, tmax, [] if args.mode == "pipeline" else [1.0, 2.0, 3.0], "test", algo, counts[4], h2d)
        print(json.dumps({"counts": counts, "tmax": tmax, "line": res, "span": [plan["lo"], plan["lo"] + plan["n"]]}))
    dist.barrier()
    dist.destroy_process_group()
""")


def run_ranks(tmp_path, extra, port, world=2):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)] + extra,
                        capture_output=True, timeout=300, env=env)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    line = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def oracle_counts(n, stride=152):
    """kept and bases kept of reads [0, n) of bench.py's fixed-length job, generated in ONE piece."""
    import torch
    import oracle_bind as ob
    import bench
    qual = bench.wl.se_shard(torch, torch.device("cpu"), bench.SEED, 0, n, 150, stride)
    cuts, _ = ob.oracle_trim_batch(ob.make_params("sanger"), qual.numpy().reshape(-1), stride=stride, read_len=150, n_reads=n)
    kept = int((cuts[:, 1] >= 0).sum())
    return kept, int(np.clip(cuts[:, 1] - cuts[:, 0], 0, None).sum())


def test_two_ranks_strong_scaling_line(tmp_path):
    """bench.py --total-reads: the job's reads split into contiguous shards (BASELINE configs[3] shape), the
    counters of the whole job on rank 0's line -- the SAME kept / discarded as the job scanned in one piece (reads
    are seeded by global block, not by rank) --, throughput = all reads x steps over the slowest rank's time."""
    total = 3 * 65536 // 2 + 17  # the shard boundary falls inside a seed block
    got = run_ranks(tmp_path, ["--total-reads", str(total)], 29533)
    kept, bases = oracle_counts(total)
    assert got["counts"][:4] == [kept, total - kept, bases, total]
    assert got["tmax"] == 1.5  # max over ranks
    line = got["line"]
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["reads_in_job"] == total and line["config"]["reads_per_gpu"] == (total + 1) // 2
    assert "sharded across 2" in line["config"]["workload"] and "configs[3]" in line["config"]["workload"]
    assert abs(line["value"] - total * 3 / 1.5) < 1e-4 * line["value"]
    assert line["kept"] == kept and line["roofline"]["kernel_ms_avg"] == 2.0
    assert len(json_line(line)) < 2500


def json_line(d):
    import json
    return json.dumps(d)


def test_two_ranks_weak_scaling_line(tmp_path):
    got = run_ranks(tmp_path, ["--reads", "7001"], 29534)
    kept, _ = oracle_counts(14002)
    line = got["line"]
    assert got["span"] == [0, 7001]
    assert line["scaling"] == "weak" and line["config"]["reads_in_job"] == 14002 and line["kept"] == kept
    assert "configs[1]" in line["config"]["workload"]


def test_mixed_workload_same_totals_on_one_and_two_ranks(tmp_path):
    """--workload mixed (BASELINE configs[4]'s batch: 75-301 bp, illumina, -n): strong scaling, and the two-shard job
    keeps and discards exactly what the one-shard job does."""
    one = run_ranks(tmp_path, ["--workload", "mixed", "--mixed-reads", "90001"], 29535, world=1)
    two = run_ranks(tmp_path, ["--workload", "mixed", "--mixed-reads", "90001"], 29536, world=2)
    assert one["counts"] == two["counts"] and one["counts"][3] == 90001
    assert 0 < two["counts"][0] < 90001  # -n discards every read with an uppercase N: a real mix of kept and discarded
    line = two["line"]
    assert line["scaling"] == "strong" and "configs[4]" in line["config"]["workload"] and line["config"]["read_len"] == "75-301"
    assert two["span"] == [0, 45001]
    assert line["roofline"]["algorithmic_bytes_per_launch"] > 2 * 75 * 45001


def test_pipeline_mode_line(tmp_path):
    """--mode pipeline (configs[3]'s async batch pipeline): per-rank and total H2D rates on the line, the step named
    as PCIe-inclusive."""
    got = run_ranks(tmp_path, ["--mode", "pipeline", "--total-reads", "20001", "--batches", "8"], 29537)
    line = got["line"]
    assert line["config"]["mode"] == "pipeline" and "async batch pipeline: 8 batches per rank" in line["config"]["workload"]
    assert line["h2d_GBps_per_rank"] == [10.0, 11.0] and line["h2d_GBps_total"] == 21.0
    assert "PCIe" in line["metric"] and line["roofline"]["traffic"] is None
