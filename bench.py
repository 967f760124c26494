"""Benchmark of the keypoint-training hot path (BASELINE.json metric: images/sec, keypoint fwd+bwd(+Adam) @512x512,
per-GPU batch 32, bf16 storage / f32 accumulate), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL. Either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or - when
WORLD_SIZE is unset - this process becomes a LAUNCHER: it never touches the GPU, checks that N devices are visible,
starts the N ranks as child processes of this same script with the rendezvous variables set (127.0.0.1, a free port),
relays rank 0's single JSON line and exits with the ranks' exit code (any rank failing ends the others).

A step = forward + losses + backward + gradient all-reduce (N>1) + Adam + weight repack on one synthetic batch
that is resident in HBM before the timed region. Rank 0 prints ONE JSON line.

`--dry-run-cpu` (tests): the same launcher, rendezvous, two-phase gradient exchange, barrier and max-over-ranks timing
on CPU tensors over gloo - no kernel runs, the line says "dry_run": true and is not a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
GMAC_PER_IMAGE = 17.946     # forward MACs per 512x512 image


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_commands(n, argv, port, base_env=None, shared_device=False):
    """[(argv, env)] of the N rank processes the launcher starts: this script again with the caller's arguments and the
    rendezvous variables torch.distributed.run would set (shared_device: every rank on device 0 - the rehearsal mode)."""
    base_env = dict(os.environ if base_env is None else base_env)
    out = []
    for r in range(n):
        env = dict(base_env, RANK=str(r), LOCAL_RANK="0" if shared_device else str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out.append(([sys.executable, os.path.abspath(__file__)] + list(argv), env))
    return out


def visible_gpu_count(kfd_root="/sys/class/kfd/kfd/topology/nodes", env=None):
    """GPUs this process' children will see, counted WITHOUT the HIP / HSA runtime: the kfd topology nodes that have SIMDs
    (CPU nodes have simd_count 0), narrowed by the *_VISIBLE_DEVICES lists the runtime honours. 0 when there is no kfd
    topology at all (without the kfd driver ROCm offers no device either)."""
    env = os.environ if env is None else env
    try:
        nodes = sorted(os.listdir(kfd_root), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return 0
    have = 0
    for nd in nodes:
        try:
            with open(os.path.join(kfd_root, nd, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
        except OSError:
            continue              # a node this user may not read is a device the runtime will not offer either
        if int(props.get("simd_count", "0")) > 0:
            have += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def launch_ranks(n, argv, dry_run=False, timeout_s=1500.0, shared_device=False):
    """The launcher (parent of the ranks). Makes no GPU call: devices are counted from the kfd topology in sysfs, never
    through torch / HIP (torch's own count falls back to hipGetDeviceCount when amdsmi is missing, which would bring
    the runtime up in a process that forks)."""
    if not dry_run:
        have = visible_gpu_count()
        if have < (1 if shared_device else n):
            sys.stderr.write(f"bench.py: --gpus {n} needs {n} devices, {have} visible\n")
            return 2
    cmds = rank_commands(n, argv, _free_port(), shared_device=shared_device)
    procs = []
    import tempfile
    out0 = tempfile.TemporaryFile()      # (a file, not a pipe: nobody reads while the ranks run, and a full pipe would block rank 0)
    try:
        for r, (cmd, env) in enumerate(cmds):
            # rank 0's stdout carries the JSON line; the other ranks print nothing on stdout
            procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=out0 if r == 0 else subprocess.DEVNULL))
        deadline = time.monotonic() + timeout_s
        rcs = [None] * n
        line = None
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            if any(rc not in (None, 0) for rc in rcs):
                break                     # a rank died: its peers would wait in a collective for ever
            if time.monotonic() > deadline:
                sys.stderr.write(f"bench.py: ranks still running after {timeout_s:.0f} s\n")
                break
            time.sleep(0.05)
        if rcs[0] == 0:
            out0.seek(0)
            line = out0.read().decode()
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
    if any(rc != 0 for rc in rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return next((rc for rc in rcs if rc is not None and rc > 0), 1)   # the failing rank's own code, not a peer's SIGTERM
    # exactly ONE line on stdout: rank 0's JSON; anything else a library printed there goes to stderr
    lines = [ln for ln in (line or "").splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if not js or ln is not js[-1]:
            sys.stderr.write(ln + "\n")
    if not js:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    sys.stdout.write(js[-1] + "\n")
    sys.stdout.flush()
    return 0


def dry_run_cpu(args):
    """One rank of the CPU rehearsal: rendezvous over gloo, the Trainer's two-phase exchange on an arena-sized CPU tensor,
    the bench's barrier / max-over-ranks timing and JSON line. Nothing here is a measurement of the hot path."""
    from multiposenet_amd import net as mnet
    from multiposenet_amd.parallel import GradientAllReducer, init_distributed
    rank, local_rank, world = init_distributed("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("MPN_BENCH_FAIL_RANK") == str(rank):       # tests: a rank that dies before the first collective
        sys.exit(7)
    shapes = {k: v for k, v in mnet.variable_shapes(1.0).items() if mnet.is_trainable(k)}
    arena = mnet._Arena(shapes, "cpu")
    grad = arena.new()
    split, deep = mnet.backbone_grad_end_of(arena), mnet.backbone_deep_begin_of(arena)
    red = GradientAllReducer(grad)
    for _ in range(args.warmup):
        red.start(split, None), red.start(deep, split), red.start(0, deep), red.finish()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    fail_at = os.environ.get("MPN_BENCH_FAIL_AT_STEP")           # tests: "<rank>:<step>" - that rank raises inside its step loop,
    for i in range(args.steps):                                   # after earlier collectives have completed; its peers block in theirs
        if fail_at is not None and fail_at == f"{rank}:{i}":
            raise RuntimeError(f"rank {rank}: injected failure in step {i}")
        grad.fill_(float(rank + 1))
        red.start(split, None)       # head end first (overlaps the backbone's backward on the GPU path)
        red.start(deep, split)       # the deep backbone blocks (overlaps the shallow blocks' backward)
        red.start(0, deep)
        red.finish()
    torch.distributed.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    # per-rank step times to rank 0 (the GPU path's `ms_per_step_per_rank`), then the max over ranks
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    torch.distributed.all_gather(per_rank, t)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    want = world * (world + 1) / 2.0
    okt = torch.tensor([1 if bool((grad == want).all()) else 0], dtype=torch.int32)
    torch.distributed.all_reduce(okt, op=torch.distributed.ReduceOp.MIN)      # the sum is right on EVERY rank
    ok = bool(okt.item())
    if rank == 0:
        print(json.dumps({"metric": "images/sec keypoint fwd+bwd+Adam @512x512 bs32/GPU", "value": None, "unit": "images/s",
                          "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * float(t) / max(1, args.steps), 3), "scaling": "weak",
                          "config": {"rccl_ranks": torch.distributed.get_world_size(), "backend": "gloo",
                                     "gradient_elements": grad.numel(), "allreduce_sum_ok": ok,
                                     "exchange_ranges": [[split, grad.numel()], [deep, split], [0, deep]],
                                     "ms_per_step_per_rank": [round(1e3 * float(x) / max(1, args.steps), 3) for x in per_rank]}}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fuse-stem-stats", action="store_true", help="A/B: batch-norm statistics of the stem output as a separate pass")
    ap.add_argument("--no-fuse-conv-bn", action="store_true", help="A/B: batch-norm reductions behind the 3x3 data gradients as separate passes")
    ap.add_argument("--dry-run-cpu", action="store_true", help="rehearse launcher + exchange on CPU over gloo (tests)")
    ap.add_argument("--rehearse-shared-device", action="store_true",
                    help="REHEARSAL, not a measurement: the N ranks of the real GPU path (graphs, exchange, every rank-0 leg) all on "
                         "device 0 with the gradient exchange over gloo - what a one-GPU box can run of the N > 1 control flow")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    ws = os.environ.get("WORLD_SIZE")
    is_rank = ws is not None and "RANK" in os.environ and int(ws) == args.gpus
    if args.gpus > 1 and not is_rank:
        if ws is not None and "RANK" in os.environ:
            # started by an outer launcher whose world is not the one asked for: refuse rather than run fewer ranks silently
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but the environment says WORLD_SIZE={ws}\n")
            sys.exit(2)
        # a WORLD_SIZE some scheduler exported without RANK is not a rank's environment: launch our own ranks
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], dry_run=args.dry_run_cpu,
                              shared_device=args.rehearse_shared_device))                # before anything touches the GPU
    if args.dry_run_cpu:
        sys.exit(dry_run_cpu(args))

    from multiposenet_amd import _lib
    from multiposenet_amd.parallel import init_distributed
    rank, local_rank, world = init_distributed("gloo" if args.rehearse_shared_device else "nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    _lib.lib()   # fail loudly if the HIP library is missing

    from multiposenet_amd.net import KeypointNet
    from multiposenet_amd.synthetic import synthetic_batch
    from multiposenet_amd.train import Trainer
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = f"cuda:{local_rank}"
    net = KeypointNet(dtype=dt, device=dev, seed=0)          # identical replicas on every rank
    if args.no_fuse_conv_bn:
        net.fuse_conv_bn = False
    if args.no_fuse_stem_stats:
        net.fuse_stem_stats = False
    params = {"initial_learning_rate": 3e-4, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    force_dp = os.environ.get("MPN_DP_FORCE_COLLECTIVE", "0") == "1"   # rehearse the data-parallel path with one rank
    trainer = Trainer(net, params, use_graph=not args.no_graph, distributed=world > 1 or force_dp)
    feats, labels = synthetic_batch(args.batch, args.size, args.size, rank=rank, device=dev)
    # the synthetic batch lives in the trainer's own input buffers (resident in HBM before the timed region): what a
    # device-side loader does; step() then has nothing to copy
    feats, labels = trainer.input_buffers(feats, labels)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(feats, labels)
    # per-step HIP events on the launch stream (graph replays and the eager path both run on torch's current stream):
    # the distribution of the step time next to the mean that the wall clock gives
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if trainer.reducer is not None:
        trainer.measure_comm = True
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        trainer.step(feats, labels)
        ev[i + 1].record()
    barrier()
    dt_s = time.perf_counter() - t0
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))

    def pct(q):
        return round(step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))], 3)
    per_rank_ms = [round(1e3 * dt_s / args.steps, 3)]
    if world > 1:
        mine = torch.tensor([dt_s], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)                 # a straggler shows in the line, not only in the maximum
        per_rank_ms = [round(1e3 * float(x.item()) / args.steps, 3) for x in every]
        t = mine.clone()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt_s = float(t.item())
    loss = float(trainer._losses[6])
    images_per_s = args.batch * world * args.steps / dt_s
    out = {
        "metric": "images/sec keypoint fwd+bwd+Adam @512x512 bs32/GPU", "value": round(images_per_s, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt_s / args.steps, 3),
        "ms_per_step_events": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "min": round(step_ms[0], 3),
                               "max": round(step_ms[-1], 3)},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"MobileNet-v1+FPN+keypoint_subnet fwd+bwd+Adam, {args.size}x{args.size}, per-GPU batch {args.batch}",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                   "final_total_loss": loss,
                   "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   # (in the headline, not only in a later key) what this build's numbers mean for north_star's parity bar
                   "parity": ("bf16 storage / f32 accumulate: heatmap arg-max identical to the f32 build on every decided channel "
                              "(bf16_vs_f32_trained); the 1e-3 / identical-arg-max bar against the f64 oracle is met by the f32 build "
                              "(f32_build: ~4.9x slower)") if args.dtype == "bf16" else
                             "f32 build: logits within 1e-3 of the f64 oracle, arg-max identical on every decided channel",
                   "ms_per_step_per_rank": per_rank_ms},
    }
    if trainer.reducer is not None and trainer.comm_events:
        # exposed (not overlapped) gradient exchange per step: from the end of the backbone's backward graph to the moment
        # the launch stream may run the optimizer graph (rank 0's view)
        ex = sorted(a.elapsed_time(b) for a, b in trainer.comm_events[-args.steps:])
        out["config"]["exposed_allreduce_ms_per_step"] = {"median": round(ex[len(ex) // 2], 3), "max": round(ex[-1], 3)}
    if rank == 0 and not args.no_roofline:
        from bench_legs import dominant_kernel_roofline, whole_step_mfma_fraction
        out["roofline"] = dominant_kernel_roofline(net, args.batch, args.size, dt, b=trainer._static_bufs())
        out["config"]["step_mfma_frac_of_peak"] = whole_step_mfma_fraction(args.batch, args.size, dt_s / args.steps)
        from bench_legs import conv_wgrad_roofline
        out["roofline_wgrad"] = conv_wgrad_roofline(net, args.batch, args.size, dt, b=trainer._static_bufs())   # the largest family by time
        if dt == torch.bfloat16:
            from bench_legs import in_step_families, mfma_achievable_peak
            ceil = mfma_achievable_peak()     # what a hipcc-scheduled MFMA stream sustains on this box
            if ceil is not None:
                for r in (out["roofline"], out["roofline_wgrad"]):
                    r["achievable_peak"] = ceil["with_lds_operand_reads_TFLOPs"]
                    r["frac_of_achievable"] = round(r["achieved"] / ceil["with_lds_operand_reads_TFLOPs"], 4)
                out["roofline"]["achievable_peak_detail"] = ceil
            # the north star's kernel families inside the real step (per-launch HIP events over one eager step). N = 1 only: it
            # STEPS the trainer, and with more ranks a step contains the gradient exchange - rank 0 must not enter a collective alone
            if world == 1:
                out["north_star_in_step"] = in_step_families(trainer, feats, labels, args.batch, args.size)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from bench_legs import cpu_baseline, decode_benchmark, render_benchmark, prn_benchmark
        from bench_legs import host_fed_rate
        out["host_fed"] = host_fed_rate(trainer, feats, labels)     # PCIe-inclusive rate (reported beside `value`, never as it)
        out["cpu_baseline"] = cpu_baseline(args.size)
        out["decode"] = decode_benchmark(32)
        out["label_render"] = render_benchmark(args.batch, args.size, args.size)
        del trainer, net
        torch.cuda.empty_cache()
        if args.batch == 32 and args.size == 512 and dt == torch.bfloat16:
            from bench_legs import north_star_kernels, f32_build_rate, bf16_vs_f32_argmax_agreement
            out["north_star_kernels"] = north_star_kernels(args.batch)
            out["f32_build"] = f32_build_rate(args.batch, args.size)          # the build that meets the 1e-3 parity bound
            out["bf16_vs_f32"] = bf16_vs_f32_argmax_agreement(args.batch, args.size)
            from bench_legs import trained_bf16_parity
            tp = trained_bf16_parity()                                         # ... and on TRAINED (sharp) heatmaps
            out["bf16_vs_f32_trained"] = {k: v for k, v in tp.items() if not k.startswith("_")}
        out["prn"] = prn_benchmark(128)
        from bench_legs import retinanet_benchmark
        out["retinanet"] = retinanet_benchmark(16)               # BASELINE config 4
        from bench_legs import joint_inference_benchmark
        out["joint_inference"] = joint_inference_benchmark()     # create_pb.py's graph behind inference/detector.py
    if args.rehearse_shared_device:
        out["rehearsal"] = f"{world} ranks share ONE device, gradient exchange over gloo through the host: control flow only, not a measurement"
        out["value"] = None
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():   # (also the 1-rank rehearsal, MPN_DP_FORCE_COLLECTIVE=1)
        torch.distributed.barrier()   # rank 0 runs the roofline leg after the timed region: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
