"""CPU: `python bench.py --gpus N` launches its own N ranks. The launcher never touches the GPU;
the `--dry-run-cpu` mode runs rendezvous + the Trainer's two-phase gradient exchange + barrier / max-over-ranks timing
over gloo through the same launcher, so command construction, env plumbing, JSON relay and failure handling are covered
without a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_rank_commands_carry_the_rendezvous_variables():
    import bench
    cmds = bench.rank_commands(4, ["--gpus", "4", "--steps", "3"], 12345, base_env={"PATH": "/usr/bin"})
    assert len(cmds) == 4
    for r, (argv, env) in enumerate(cmds):
        assert argv[0] == sys.executable and argv[1] == BENCH and argv[2:] == ["--gpus", "4", "--steps", "3"]
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "12345"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"


def test_devices_are_counted_from_the_kfd_topology_without_the_runtime(tmp_path):
    """ADVICE r3 / VERDICT r3 item 6: the launcher counts GPUs from sysfs (nodes with SIMDs), narrowed by the
    *_VISIBLE_DEVICES lists - torch.cuda / HIP are never asked."""
    import bench
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):          # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    root = str(tmp_path)
    assert bench.visible_gpu_count(root, env={}) == 3
    assert bench.visible_gpu_count(root, env={"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert bench.visible_gpu_count(root, env={"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}) == 1
    assert bench.visible_gpu_count(root, env={"CUDA_VISIBLE_DEVICES": ""}) == 0
    assert bench.visible_gpu_count(str(tmp_path / "absent"), env={}) == 0
    import inspect
    assert "device_count" not in inspect.getsource(bench.launch_ranks)


def test_a_world_size_without_a_rank_is_not_mistaken_for_a_rank():
    """A scheduler that exports WORLD_SIZE alone must not turn `--gpus 2` into a silent single-rank run."""
    r = _run(["--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2"}, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2


def test_gpus_2_without_two_devices_exits_nonzero_in_seconds():
    import bench
    if bench.visible_gpu_count() >= 2:
        import pytest
        pytest.skip("two devices visible: the launcher would start a real run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=120)
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr
    assert r.stdout.strip() == ""


def test_two_rank_gloo_dry_run_through_the_launcher_prints_one_json_line():
    r = _run(["--gpus", "2", "--dry-run-cpu", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["rccl_ranks"] == 2 and out["config"]["allreduce_sum_ok"] is True
    assert out["config"]["gradient_elements"] >= 5521490


def test_eight_rank_gloo_dry_run_through_the_launcher():
    """no 8-GPU node was available to any round of this build, so the launcher, the rendezvous, the three-range
    gradient exchange (head end, deep backbone blocks, the rest: together the whole arena, once), the per-rank timing gather and
    the max-over-ranks line run at the REAL rank count of BASELINE config 3 over gloo: every rank ends with the sum 1 + ... + 8."""
    r = _run(["--gpus", "8", "--dry-run-cpu", "--steps", "3", "--warmup", "1"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["dry_run"] is True and out["n_gpus"] == 8 and cfg["rccl_ranks"] == 8 and cfg["allreduce_sum_ok"] is True
    assert len(cfg["ms_per_step_per_rank"]) == 8 and out["ms_per_step"] >= max(cfg["ms_per_step_per_rank"]) - 1e-3
    (a0, a1), (b0, b1), (c0, c1) = cfg["exchange_ranges"]
    assert c0 == 0 and c1 == b0 and b1 == a0 and a1 == cfg["gradient_elements"]      # the ranges tile the arena in backward order


def test_a_rank_dying_among_eight_ends_the_other_seven():
    r = _run(["--gpus", "8", "--dry-run-cpu", "--steps", "4", "--warmup", "0"], env={"MPN_BENCH_FAIL_AT_STEP": "5:2"}, timeout=600)
    assert r.returncode not in (0, None) and r.returncode > 0
    assert "injected failure in step 2" in r.stderr and "rank exit codes" in r.stderr
    assert r.stdout.strip() == ""


def test_a_dying_rank_ends_the_others_and_the_launcher_returns_its_code():
    r = _run(["--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "0"], env={"MPN_BENCH_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 7
    assert "rank exit codes" in r.stderr
    assert r.stdout.strip() == ""


def test_a_rank_raising_inside_its_step_loop_ends_its_peers_within_seconds():
    """not before the first collective (above) but in the middle of the timed loop, when the peers are
    already blocked in an all-reduce that will never complete: the launcher sees the exit code and ends them."""
    import time
    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--dry-run-cpu", "--steps", "6", "--warmup", "1"], env={"MPN_BENCH_FAIL_AT_STEP": "1:3"}, timeout=180)
    took = time.monotonic() - t0
    assert r.returncode not in (0, None) and r.returncode > 0
    assert "injected failure in step 3" in r.stderr and "rank exit codes" in r.stderr
    assert r.stdout.strip() == ""
    assert took < 90, took            # (two interpreter + torch start-ups; the wait for the dead rank itself is the 50 ms poll)


def test_ranks_started_by_an_outer_launcher_are_not_relaunched():
    """WORLD_SIZE in the environment (torch.distributed.run) = this process IS a rank: a mismatch with --gpus asserts."""
    r = _run(["--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "0"],
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_rank0_only_legs_never_enter_the_gradient_exchange(monkeypatch):
    """After the timed region rank 0 alone runs the roofline legs. `in_step_families` STEPS the trainer: with more than one
    rank a step contains the all-reduce, and a collective entered by one rank hangs the job at the final barrier. The leg
    detaches the reducer for its steps (and restores it), and bench.py runs it at N = 1 only."""
    import inspect
    import torch
    import bench, bench_legs

    class Reducer:
        grad_scale = 0.5

        def all_reduce(self):
            raise AssertionError("a collective on one rank")
        start = finish = all_reduce

    class StubNet:
        """the state the leg snapshots before its three profiled steps and puts back afterwards"""
        def __init__(self):
            self.theta, self.adam_m, self.adam_v = torch.ones(4), torch.zeros(4), torch.zeros(4)
            self.moving, self.global_step = torch.ones(2), torch.zeros(1, dtype=torch.int64)
            self.repacked = self.marked = 0

        def repack_weights(self):
            self.repacked += 1

        def mark_variables_changed(self):
            self.marked += 1

    class StubTrainer:
        def __init__(self):
            self.use_graph, self.reducer, self.seen, self.net = True, Reducer(), [], StubNet()

        def step(self, feats, labels):
            self.seen.append((self.use_graph, self.reducer))
            self.net.theta += 1.0            # (a step moves the variables and the step counter)
            self.net.global_step += 1
            if self.reducer is not None:
                self.reducer.all_reduce()

    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    tr = StubTrainer()
    out = bench_legs.in_step_families(tr, None, None, 32, 512)
    assert tr.seen == [(False, None)] * 3 and tr.use_graph is True and isinstance(tr.reducer, Reducer)
    assert out["launches"] == 0
    # the profiled steps are measurement, not training: the model is the one the timed region left
    assert torch.equal(tr.net.theta, torch.ones(4)) and int(tr.net.global_step) == 0 and tr.net.repacked == 1 and tr.net.marked == 1
    src = inspect.getsource(bench.main)
    call = src.index("in_step_families(trainer")
    assert "if world == 1:" in src[call - 200:call]
