"""bench.py's own launcher and rank plumbing on CPU: `--gpus 2` with no launcher around it must start two ranks itself,
run the timed-region protocol with a collective inside every repetition, and hand back ONE JSON line.  `--dry-run` steps a
small band per rank with the CPU restatement over gloo (no GPU here); the GPU run shares everything but the stepping."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_two_ranks_dry_run(oracle):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "6", "--warmup", "2",
                        "--reps", "3"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # rank 0's line, relayed by the parent, and nothing else
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2
    assert line["rccl"]["world"] == 2 and line["rccl"]["nranks_seen"] == 2.0
    assert line["rccl"]["reductions_per_timed_repetition"] == 2          # 6 steps in launches of 4: one full, one partial
    assert line["repetitions"]["n"] == 3 and line["repetitions"]["min"] <= line["ms_per_step"] <= line["repetitions"]["max"]
    assert line["counters"]["particles"] == 2 * 48 * 48 and line["barrier"].startswith("shared-memory epoch barrier")
    assert line["value"] > 0 and line["unit"] == "particle-steps/s"
    # the legs a scaling sweep of the driver's command carries beside the headline: config 5 (strong scaling, packed ring) and
    # the frame loop of a row-band job with the draw's exchange inside
    c3s = line["c3_strong"]            # the metric's own particles as ONE texture over the ranks
    assert c3s["scaling"] == "strong" and c3s["n_gpus"] == 2 and c3s["particles"] == 2 * c3s["particles_per_gpu"] and c3s["rows_per_gpu"] == 32
    assert c3s["rccl"]["nranks_seen"] == 2.0 and c3s["value"] > 0
    assert line["legs_failed"] == []
    c5 = line["c5"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == 2 and c5["particles"] == 2 * c5["particles_per_gpu"] == 32 * 64
    assert c5["rccl"]["nranks_seen"] == 2.0 and c5["rccl"]["reductions"] == 2 and c5["value"] > 0 and "roofline" in c5
    fls = line["frame_loop_sharded"]
    assert fls["n_gpus"] == 2 and fls["frames"] == 2 and fls["wall_ms_per_frame"] > 0
    assert fls["sent_bytes_per_draw"] > 0 and fls["received_bytes_per_draw"] > 0 and fls["fragments_per_draw_all_ranks"] > 100


def test_a_rank_without_shared_memory_sends_every_rank_to_the_torch_barrier(oracle):
    """the bracket's barrier is the ranks meeting in shared memory - every rank or none: one that cannot attach the segment takes
    all of them to torch.distributed's barrier, and nobody is left inside a collective the others skipped"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TH_BENCH_TEST_NOSHM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1",
                        "--reps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert line["barrier"] == "torch.distributed barrier" and line["value"] > 0 and line["legs_failed"] == []


def test_child_failure_is_the_parents_status():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--config", "nope"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip()


def test_roofline_fraction_names_its_bound():
    """`frac` is the fraction of the bound the entry names and never the equivalent bandwidth of a fused launch."""
    sys.path.insert(0, ROOT)
    import bench

    class J:
        particles_rank = 1 << 24
    # fused flow-only launch: equivalent bandwidth 2.2 x the peak; counters say VALU 0.34, physical HBM 0.19
    c = {"FETCH_SIZE": 300e3, "WRITE_SIZE": 200e3, "SQ_INSTS_VALU": 2.55e8}
    e = bench.bind(bench.roofline_entry(J, 0.61e-3, 20, c, 32), True)
    assert e["equivalent_frac"] > 1 and e["bound"] == "valu" and 0 < e["frac"] < 1 and e["frac"] == e["valu"]["frac"]
    # the same launch when its bytes dominate
    c = {"FETCH_SIZE": 1.2e6, "WRITE_SIZE": 1.0e6, "SQ_INSTS_VALU": 1e7}
    e = bench.bind(bench.roofline_entry(J, 0.61e-3, 20, c, 32), True)
    assert e["bound"] == "hbm" and e["frac"] == e["hbm_physical"]["frac"] < 1
    # no counters: unknown, not a number above 1
    e = bench.bind(bench.roofline_entry(J, 0.61e-3, 20, None, 32), True)
    assert e["frac"] is None and e["equivalent_frac"] > 1
    # one step per launch streams its algorithmic bytes
    e = bench.bind(bench.roofline_entry(J, 0.134e-3, 1, None, 32), False)
    assert e["bound"] == "hbm" and e["frac"] == e["equivalent_frac"] < 1


def test_a_leg_that_never_comes_back_does_not_cost_the_line(oracle):
    """Rank 1 never reaches the config-5 leg's first collective: after the leg's deadline rank 0 prints the line it has - the
    headline, the leg marked as abandoned, the leg behind it as not run - and every rank ends (benchlib/sidelegs.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TH_BENCH_TEST_HANG="c5:1", TH_BENCH_LEG_TIMEOUT="8")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1",
                        "--reps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 2 and line["rccl"]["nranks_seen"] == 2.0
    assert "no result within 8 s" in line["c5"]["error"] and "c5" in line["frame_loop_sharded"]["skipped"]
    assert line["legs_failed"] == ["c5"] and line["c3_strong"]["value"] > 0          # (rc 0 keeps the headline; THIS names what hung)


def test_a_communicator_that_never_comes_up_ends_in_a_fresh_child_with_the_line(oracle):
    """Rank 1 never reaches the communicator's start: the others stand inside a collective nobody can call them back from.
    After TH_BENCH_COMM_TIMEOUT every rank starts the bench again as a child process with --no-library-comm (a rendezvous of
    its own) and ends with its status; rank 0's child prints the line, which says why - well inside the job's own deadline."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TH_BENCH_TEST_HANG="comm_init:1", TH_BENCH_COMM_TIMEOUT="6", TH_BENCH_TIMEOUT="400")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1",
                        "--reps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 300
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 2 and line["rccl"]["nranks_seen"] == 2.0
    assert "had not come up on every rank within 6 s" in line["rccl"]["fallback"]
    assert line["legs_failed"] == [] and line["c5"]["value"] > 0
    assert "starts again as a child process" in r.stderr


def test_every_deadline_lies_under_the_drivers():
    """the driver gives a bench run 1800 s (BENCH_r05.json: timeout_s): the job's own end, the legs' and the communicator's
    deadlines all come before it, and the legs are cut to what is left of the job's time"""
    sys.path.insert(0, ROOT)
    from benchlib import sidelegs
    for k in ("TH_BENCH_TIMEOUT", "TH_BENCH_LEG_TIMEOUT", "TH_BENCH_COMM_TIMEOUT"):
        assert k not in os.environ
    assert sidelegs.job_seconds() <= 1500
    legs = sidelegs.SideLegs({}, 0)
    assert legs.limit <= 300
    assert sidelegs.comm_deadline(0, [], "bench.py").seconds <= 300
    keep = sidelegs.T_START
    try:
        sidelegs.T_START = keep - (sidelegs.job_seconds() - 100.0)      # 100 s of the job's time left
        legs.run("quick", lambda: {"ok": 1})
        assert legs.seconds <= 100.0 and legs.line["quick"] == {"ok": 1} and legs.line["legs_failed"] == []
    finally:
        sidelegs.T_START = keep
