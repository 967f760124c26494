"""GPU: the data-parallel train step (one process per device, RCCL all-reduce) against the single-device step.

`test_two_rank_rccl_step_equals_single_rank_step` needs >= 2 visible devices and skips otherwise (a one-GPU box):
two ranks train on IDENTICAL batches, so the all-reduced gradient sum is exactly twice each rank's gradient and the
averaged step must reproduce the single-rank step bit for bit - variables, Adam slots and (per-replica) moving statistics.
`test_one_rank_rccl_rehearsal...` runs everywhere: the three-graph step + the real RCCL collectives with world size 1.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_steps(device, distributed, use_graph=True, steps=2):
    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd.synthetic import synthetic_batch
    from multiposenet_amd.train import Trainer
    net = KeypointNet(dtype=torch.bfloat16, device=device, seed=0)
    tr = Trainer(net, HP, use_graph=use_graph, distributed=distributed)
    feats, labels = synthetic_batch(2, 128, 128, rank=0, device=device)      # the SAME batch on every rank
    losses = [tr.step(feats, labels).cpu().numpy().copy() for _ in range(steps)]
    torch.cuda.synchronize(device)
    return {"losses": np.stack(losses), "theta": net.theta.cpu().numpy(), "m": net.adam_m.cpu().numpy(),
            "v": net.adam_v.cpu().numpy(), "moving": net.moving.cpu().numpy(), "grad": net.grad.cpu().numpy()}


def _worker_main():
    """Entry point of a spawned rank (python tests/test_dp_gpu.py <out_prefix>); env: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*."""
    sys.path.insert(0, ROOT)
    from multiposenet_amd.parallel import init_distributed
    # MPN_TEST_BACKEND=gloo: two ranks that SHARE one device (a 1-GPU box: RCCL refuses two ranks on one GPU, gloo moves the
    # buckets through the host) - the same three-graph step, bucketed exchange and 1/world average with a real second rank
    backend = os.environ.get("MPN_TEST_BACKEND", "nccl")
    rank, local_rank, world = init_distributed(backend)
    torch.cuda.set_device(local_rank)
    out = _run_steps(f"cuda:{local_rank}", distributed=True)
    np.savez(f"{sys.argv[1]}.rank{rank}.npz", world=world, **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _spawn(world, prefix, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), prefix], env=env, cwd=ROOT))
    # poll all ranks: one that dies leaves its peers blocked in a collective - end them instead of waiting them out
    import time
    deadline = time.monotonic() + 300
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs) or time.monotonic() > deadline:
                break
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    rcs = [p.returncode for p in procs]
    assert rcs == [0] * world, f"rank exit codes {rcs}"
    return [dict(np.load(f"{prefix}.rank{r}.npz")) for r in range(world)]


def test_two_rank_rccl_step_equals_single_rank_step(cuda, tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible devices (one process per GPU)")
    want = _run_steps("cuda:0", distributed=False)
    got = _spawn(2, str(tmp_path / "dp"))
    for r, g in enumerate(got):
        assert int(g["world"]) == 2
        for k in ("losses", "theta", "m", "v", "moving"):
            np.testing.assert_array_equal(g[k], want[k], err_msg=f"rank {r}: {k}")
        # the arena holds the all-reduced SUM (the 1/world average is folded into the Adam kernel): exactly 2 x one rank's
        np.testing.assert_array_equal(g["grad"], 2.0 * want["grad"], err_msg=f"rank {r}: grad")


def _ranks_on_one_device_over_gloo(world, prefix):
    """`world` ranks on ONE device (every rank LOCAL_RANK 0, backend gloo): what a 1-GPU box can run of the data-parallel step with real
    peers; returns what each rank saved (tests/test_dp_gpu.py run as a rank: losses, gradient arena, variables, Adam slots, moving statistics)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world),
                   MPN_TEST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), prefix], env=env, cwd=ROOT))
    import time
    deadline = time.monotonic() + 400
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs) or time.monotonic() > deadline:
                break
            time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    assert [p.returncode for p in procs] == [0] * world
    return [dict(np.load(f"{prefix}.rank{r}.npz")) for r in range(world)]


def test_two_ranks_on_one_device_over_gloo_equal_single_rank_step(cuda, tmp_path):
    """World size 2 on ONE device (both ranks LOCAL_RANK 0, backend gloo): what a 1-GPU box can check of the data-parallel step
    with a real second rank - identical batches on both ranks, so the all-reduced sum is exactly twice each rank's gradient and
    the averaged step reproduces the single-rank step bit for bit (variables, Adam slots, per-replica moving statistics)."""
    want = _run_steps("cuda:0", distributed=False)
    for r, g in enumerate(_ranks_on_one_device_over_gloo(2, str(tmp_path / "dpg"))):
        assert int(g["world"]) == 2
        for k in ("losses", "theta", "m", "v", "moving"):
            np.testing.assert_array_equal(g[k], want[k], err_msg=f"rank {r}: {k}")
        np.testing.assert_array_equal(g["grad"], 2.0 * want["grad"], err_msg=f"rank {r}: grad")


def test_four_ranks_on_one_device_over_gloo_follow_the_single_rank_step(cuda, tmp_path):
    """Round 6 (no multi-GPU node in any round): the same with FOUR ranks on the one device - four graphs around three exchanges per step,
    the 1 / world average in the Adam kernel - the most a 1-GPU box's process guard allows next to the test runner. A sum of four equal
    f32 values is exact only when the collective adds them pairwise, so the sum is held to 4 x one rank's gradient within one f32
    rounding of a three-fold sum and the variables / Adam slots / moving statistics to that rounding carried through two steps; every
    rank must hold the SAME bits (the all-reduce leaves identical arenas everywhere)."""
    want = _run_steps("cuda:0", distributed=False)
    got = _ranks_on_one_device_over_gloo(4, str(tmp_path / "dp4"))
    for r, g in enumerate(got):
        assert int(g["world"]) == 4
        np.testing.assert_allclose(g["grad"], 4.0 * want["grad"], rtol=3e-7, atol=0, err_msg=f"rank {r}: grad")
        np.testing.assert_allclose(g["losses"], want["losses"], rtol=1e-6, err_msg=f"rank {r}: losses")
        np.testing.assert_array_equal(g["moving"], want["moving"], err_msg=f"rank {r}: moving statistics are per replica")
        for k in ("theta", "m", "v"):
            np.testing.assert_allclose(g[k], want[k], rtol=2e-5, atol=2e-7, err_msg=f"rank {r}: {k}")
        for k in ("grad", "theta", "m", "v"):
            np.testing.assert_array_equal(g[k], got[0][k], err_msg=f"rank {r} differs from rank 0: {k}")


def test_one_rank_rccl_rehearsal_equals_plain_step(cuda, tmp_path):
    """World size 1 through init_distributed + the three-graph step + RCCL all-reduce (MPN_DP_FORCE_COLLECTIVE=1): the
    same bits as the one-graph step without a process group."""
    want = _run_steps("cuda:0", distributed=False)
    got = _spawn(1, str(tmp_path / "dp1"), {"MPN_DP_FORCE_COLLECTIVE": "1"})[0]
    assert int(got["world"]) == 1
    for k in ("losses", "theta", "m", "v", "moving", "grad"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_distributed_without_process_group_is_refused(cuda, monkeypatch):
    """Never a silent world = 1: WORLD_SIZE > 1 without an initialised process group raises."""
    from multiposenet_amd.parallel import GradientAllReducer
    monkeypatch.setenv("WORLD_SIZE", "2")
    if torch.distributed.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    with pytest.raises(RuntimeError, match="not initialised"):
        GradientAllReducer(torch.zeros(16, device="cuda"))


if __name__ == "__main__":
    _worker_main()


def test_bench_two_rank_path_rehearsed_on_one_device(cuda):
    """`bench.py --gpus 2 --rehearse-shared-device`: the launcher, both ranks of the REAL GPU path (four graphs around three
    exchanges, barrier + max-over-ranks timing, per-rank step times) and every leg rank 0 runs alone after the timed region, with
    both ranks on device 0 and the exchange over gloo. A leg that stepped the trainer on rank 0 alone once entered the all-reduce
    there and would have hung the first real multi-rank run at the final barrier - this is the run that shows it."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-device", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["value"] is None and "rehearsal" in out
    assert len(out["config"]["ms_per_step_per_rank"]) == 2 and "exposed_allreduce_ms_per_step" in out["config"]
    assert "roofline" in out and "north_star_in_step" not in out
    assert np.isfinite(out["config"]["final_total_loss"])


def test_one_rank_rccl_rehearsal_of_the_bench_reports_a_small_exposed_all_reduce(cuda):
    """`MPN_DP_FORCE_COLLECTIVE=1 python bench.py` runs the data-parallel step of BASELINE config 3 - four graphs
    around three RCCL exchanges - with one rank. The part of the exchange the step cannot hide (from the end of the backbone's
    backward graph to the moment the optimizer graph may run: the last ~1.3 MB of gradients + RCCL's own launch) is printed as
    `config.exposed_allreduce_ms_per_step` (measured 0.03-0.06 ms on this pool's boxes; printed, not gated: a wall-clock bound on a shared,
    power-capped pool fails for reasons that are not correctness - only a loose sanity bound is asserted); no multi-GPU node was available
    to measure more."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MPN_DP_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    cfg = out["config"]
    assert cfg["rccl_ranks"] == 1 and "exposed_allreduce_ms_per_step" in cfg, cfg
    print("exposed all-reduce per step (ms):", cfg["exposed_allreduce_ms_per_step"], " ms per step:", out["ms_per_step"])
    med = cfg["exposed_allreduce_ms_per_step"]["median"]
    assert np.isfinite(med) and 0.0 <= med < 0.5 * out["ms_per_step"], cfg["exposed_allreduce_ms_per_step"]
