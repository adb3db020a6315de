#!/usr/bin/env python3
"""bench.py -- scored trajectories/sec of the local-planner tick on MI355X.

A "step" is one control tick over one batch of synthetic input: local costmap
binning + body-frame rollout + all critics + argmin (+ ONE small min all-reduce when
N > 1), with the cloud already resident in HBM (set_cloud is outside the timed
region) and the chosen cmd_vel delivered to the host every tick.

Workloads (BASELINE.json configs; SURVEY.md 8d scenes, ~25 % colliding trajectories):
  --gpus 1 (default)  C3: 16384 trajectories x 80 steps vs a 500k-point three-floor cloud --
                      the largest single-GPU configuration (C2 via --workload C2).
  --gpus N > 1        C4: 65536 trajectories x 50 steps vs a 100k-point cloud, STRONG scaling:
                      rank r scores the contiguous index range [65536 r/N, 65536 (r+1)/N), every
                      rank holds the cloud, one min all-reduce of (cost bits, -index) slots picks
                      the winner.  `--workload C2` with N > 1 is the weak-scaling variant (4096 per GPU).
`python bench.py --gpus N` starts its own N ranks (torch.distributed.run, before anything here
touches a GPU); launched under torch.distributed.run it reads RANK / WORLD_SIZE from the env.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed ticks (default 1000; 400 for C3/C4 on one GPU)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, choices=["C2", "C3", "C3P", "C4", "C5", "C5M", "shipped"],
                    help="default: C3 on one GPU, C4 (strong scaling) on several; shipped = only the tick latencies of the "
                         "reference's shipped configurations (55 / 275 / 2 trajectories)")
    ap.add_argument("--inputs", default="static", choices=["static", "moving"],
                    help="static (default, the headline): every tick sees the same pose / twist / cloud; moving: the robot "
                         "advances along the prune plan every tick and the cloud is replaced every --replace-every ticks "
                         "(always measured as config.moving_value on one GPU; --inputs moving makes it the headline value)")
    ap.add_argument("--replace-every", type=int, default=10)
    ap.add_argument("--marking-schedule", choices=["overlapped", "serial"], default="overlapped",
                    help="C5M: tick_begin -> marking update -> tick_end (the update on its own stream next to the tick's kernels, "
                         "as the reference's perception and planner threads run side by side) or feed -> update -> tick one after the other")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / moving / shipped-latency measurements")
    ap.add_argument("--scene-layout", default=None, choices=["r02", "r01"],
                    help="r02 (default): ~25 %% colliding trajectories as SURVEY 8d specifies; r01: the round-1 scenes (69-86 %%)")
    ap.add_argument("--backend", default=os.environ.get("DDDMR_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--reduce", default="auto", choices=["auto", "inlib", "torch"],
                    help="who runs the per-tick min all-reduce: the library's own RCCL communicator (inlib) or "
                         "torch.distributed (torch); auto = inlib with --backend nccl when it initialises, else torch")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="with --backend nccl: fall back to gloo if RCCL cannot initialise instead of failing")
    ap.add_argument("--contexts", type=int, default=1,
                    help="independent planner contexts (robots) per GPU, one tick in flight each; 1 = sequential ticks (the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-ceiling", action="store_true", help="skip the stream-ceiling measurement")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N`: become the launcher of N ranks.  Nothing in this process has
    imported torch or touched the GPU, and it never execs: the ranks are child processes."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # stdout carries exactly ONE JSON line (rank 0's); whatever else the ranks or their libraries
    # print there (gloo's connection banners, for one) is passed on to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:
        if line.startswith("{") and line.rstrip().endswith("}"):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    return rc


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else 0.0


def moving_inputs(scenes, sc, n_ticks):
    """Pose / twist per tick for --inputs moving: the robot follows the prune plan (one 5 cm pose every two ticks,
    20 ... 60 of its 80 poses, then back), its twist wanders inside the dynamic window, and two variants of the
    cloud alternate (the second with every 16th point displaced by a few centimetres, as a new scan would)."""
    import math
    import numpy as np
    ticks = []
    for k in range(n_ticks):
        i = 20 + ((k // 2) % 40)
        pose = tuple(sc.plan[i])
        twist = (0.5 + 0.15 * math.sin(k / 7.0), 0.05 * math.sin(k / 3.0), 0.25 * math.sin(k / 5.0))
        ticks.append(scenes.tick_input(pose=pose, twist=twist))
    rng = np.random.default_rng(1234)
    alt = sc.cloud.copy()
    alt[::16, :3] += rng.uniform(-0.03, 0.03, size=alt[::16, :3].shape).astype(np.float32)
    return ticks, [sc.cloud, alt]


def shipped_latencies(np, scenes, configs, LocalPlanner, gpu):
    """Tick latency (host wall time of dddmr_rollout_tick, result delivered) of the configurations the reference ships,
    against the C1 cloud: warm = back to back, cold = after 100 ms of idling (a 10 Hz control loop)."""
    c1 = scenes.bench_scene("C1")
    pg = scenes.playground_scene()
    cases = [("playground differential_drive_simple (55 trajectories x 21-61 steps)", pg.theory, pg.plan, pg.tick),
             ("omni_drive_simple 5 x 5 x 11 (275)", configs.omni_simple_shipped(), c1.plan, scenes.tick_input(twist=(0.3, 0.0, 0.0))),
             ("differential_drive_rotate_inplace (2 x 126 steps)", configs.rotate_inplace_shipped(), c1.plan, scenes.tick_input(twist=(0.0, 0.0, 0.0)))]
    out = []
    for label, th, plan, tick in cases:
        with LocalPlanner([th], device=gpu, max_points=len(c1.cloud)) as lp:
            lp.set_cloud(c1.cloud)
            lp.setPlan(plan)
            name = th.name.decode()
            for _ in range(20):
                r = lp.tick(name, tick)
            warm = []
            for _ in range(200):
                t0 = time.perf_counter()
                r = lp.tick(name, tick)
                warm.append(time.perf_counter() - t0)
            cold = []
            for _ in range(12):
                time.sleep(0.1)
                t0 = time.perf_counter()
                r = lp.tick(name, tick)
                cold.append(time.perf_counter() - t0)
            out.append({"theory": label, "n_trajectories": int(r.n_samples), "planner_state": int(r.planner_state),
                        "warm_tick_ms": round(median(warm) * 1e3, 4), "cold_tick_ms": round(median(cold) * 1e3, 4),
                        "cold_tick_ms_max": round(max(cold) * 1e3, 4)})
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line: the JSON.  Whatever libraries print there on the way (RCCL writes a version
    # banner on its first communicator, gloo its connection report) is sent to stderr instead.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from dddmr_navigation_amd import scenes, configs, sharding, _capi as K
    from dddmr_navigation_amd.local_planner import LocalPlanner, RolloutError

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rollout engine has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if world > n_dev and args.backend == "nccl":
        raise SystemExit(f"--gpus {world} with --backend nccl needs {world} GPUs, this node shows {n_dev} "
                         "(several ranks may share a GPU only with --backend gloo, as a rehearsal)")
    gpu = local_rank % n_dev
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = args.backend
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.full((1,), rank, dtype=torch.int64, device=dev)    # fail early if RCCL cannot move 8 bytes
                dist.all_reduce(probe, op=dist.ReduceOp.MIN)
                assert int(probe.item()) == 0
            except Exception as e:
                if not args.allow_gloo_fallback:
                    raise SystemExit(f"[bench] --backend nccl: RCCL failed to initialise ({type(e).__name__}: {e}); "
                                     "pass --allow-gloo-fallback to rehearse over gloo instead")
                print(f"[bench] RCCL unavailable ({type(e).__name__}: {e}); --allow-gloo-fallback: using gloo",
                      file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        args.backend = backend
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    # ---- workload ----
    workload = args.workload or ("C3" if world == 1 else "C4")
    if workload == "shipped":
        if world > 1:
            raise SystemExit("--workload shipped is a single-GPU latency measurement")
        sh = shipped_latencies(np, scenes, configs, LocalPlanner, gpu)
        n0, t0 = sh[0]["n_trajectories"], sh[0]["warm_tick_ms"]
        json_out.write(json.dumps({
            "metric": "scored trajectories/sec (N_traj x N_steps)", "value": round(n0 / (t0 * 1e-3), 1), "unit": "trajectories/s",
            "n_gpus": 1, "steps": 200, "warmup": 20, "ms_per_step": t0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "shipped: tick latency of the reference's shipped theories against the 5000-point C1 cloud "
                                   "(value = the playground theory's warm rate)", "shipped_tick_latency": sh},
            "roofline": None, "cpu_baseline": None}) + "\n")
        json_out.flush()
        return
    layout = args.scene_layout or scenes.DEFAULT_LAYOUT
    feed = workload in ("C5", "C5M")
    base = "C2" if feed else workload
    sc = scenes.bench_scene(base, layout)
    theory = sc.theory
    scaling = "weak"
    if base == "C2" and world > 1:
        theory.linear_x_sample = 16.0 * world          # 4096 samples per GPU
    if base in ("C3", "C4") and world > 1:
        scaling = "strong"
    if args.steps is None:
        args.steps = 400 if (base in ("C3", "C3P", "C4") and world == 1) else 1000
    name = theory.name.decode()
    b = configs.BENCH[base]
    n_steps_traj = b["steps"]

    # HIP events serialise the queue around them (~3 us each): the kernels are timed on every 8th tick
    os.environ.setdefault("DDDMR_TIMING", "2")
    os.environ.setdefault("DDDMR_TIMING_EVERY", "8")
    lp = LocalPlanner([theory], device=gpu, max_points=max(len(sc.cloud), 1 << 16), max_trajectories=1 << 20,
                      rank=rank, world_size=world)
    lp.set_cloud(sc.cloud)            # inputs resident in HBM before the timed region
    lp.setPlan(sc.plan)
    scans = None
    marking = None
    if feed:
        # perception feed fused with the tick: 10 simulated 16-ring LiDAR scans of the C2
        # scene; every step = set_scan (crop + 0.1 m voxel-hash downsample on the GPU,
        # H2D of the raw scan included) + one C2 tick on the resulting cloud
        scans = [scenes.lidar_scan(sc.cloud, seed=100 + i) for i in range(10)]
        t_bs, t_gb = (0.0, 0.0, 0.5, 0, 0, 0, 1), (0.0, 0.0, 0.0, 0, 0, 0, 1)
        if workload == "C5M":
            # ... plus the global-mode marking/clearing layer on the same scan
            from dddmr_navigation_amd import marking as marking_mod
            marking = marking_mod.bench_layer(lp, sc)
    step_no = [0]

    # ---- the per-tick exchange (N > 1) ----
    # Exact argmin over the ranks: every rank contributes (bit pattern of its best cost, -index);
    # ONE min all-reduce of the 2*world-word slot vector (16 bytes per rank) hands every rank all
    # pairs, and dddmr_rollout_resolve_words picks minimum cost / highest index among equals -- the
    # reference's `<=` scan (local_planner.cpp:456-463) over the whole batch, full doubles.
    def inlib_bootstrap():
        """Bring the library's own RCCL communicator up on every rank; True if all ranks made it.
        Rank 0 draws the RCCL unique id and hands it to the others (any broadcast will do; a C++ host
        would use its own channel); an all-zero id means "rank 0 cannot" and every rank agrees."""
        uid = bytes(128)
        if rank == 0:
            try:
                uid = lp.comm_unique_id()
            except RolloutError as e:
                print(f"[bench] in-library RCCL unavailable ({e})", file=sys.stderr, flush=True)
        t = torch.tensor(list(uid), dtype=torch.uint8, device=red_dev)
        dist.broadcast(t, src=0)
        uid = bytes(t.cpu().tolist())
        up = False
        if any(uid):
            try:
                lp.comm_init(uid, rank, world)
                up = True
            except RolloutError as e:
                print(f"[bench] rank {rank}: dddmr_rollout_comm_init failed ({e})", file=sys.stderr, flush=True)
        ok = torch.tensor([1 if up else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and up:
            lp.comm_destroy()
        return int(ok.item()) == 1

    # Which exchange the TIMED loop uses.  torch (default, `auto`): the host hides the all-reduce behind the
    # next tick (throughput; winners arrive two ticks late).  inlib: k_score -> ncclAllReduce -> k_resolve on
    # the tick's stream, every tick synchronous.  With `auto` the in-library path is measured as well, AFTER the
    # timed loop and under a watchdog (a collective that never returns must not cost the run its line).
    reduce_mode = "none"
    if world > 1:
        reduce_mode = "torch"
        if args.reduce == "inlib":
            if args.backend != "nccl":
                raise SystemExit("--reduce inlib needs --backend nccl (RCCL)")
            if not inlib_bootstrap():
                raise SystemExit("[bench] --reduce inlib: the library's RCCL communicator did not come up on every rank")
            reduce_mode = "inlib"

    # torch path: everything the host does for the all-reduce -- collecting the reduce of tick i-2,
    # issuing the one of tick i-1 -- happens between tick_begin(i) and tick_end(i), i.e. while the
    # GPU computes tick i.  The winner of a tick is delivered two ticks later; every all-reduce
    # completes inside the timed region (fence() issues and collects the last ones).  The
    # synchronous tick -> all-reduce -> resolve latency is reported separately below.
    slot_host = [torch.full((2 * world,), sharding.INT64_MAX, dtype=torch.int64).pin_memory() for _ in range(2)]
    slot_bufs = [torch.full((2 * world,), sharding.INT64_MAX, dtype=torch.int64, device=red_dev) for _ in range(2)]
    pending = []          # [(work handle, buffer)]
    resolved = [None]
    prev_words = [None]   # words of the last finished tick, not yet handed to the all-reduce
    n_issued = [0]

    def collect():
        while pending:
            work, buf = pending.pop(0)
            work.wait()
            resolved[0] = lp.resolve_words(buf.tolist())

    def issue():
        if prev_words[0] is None:
            return
        j = n_issued[0] % 2
        n_issued[0] += 1
        h = slot_host[j]
        h[2 * rank] = prev_words[0][0]
        h[2 * rank + 1] = prev_words[0][1]
        prev_words[0] = None
        slot_bufs[j].copy_(h, non_blocking=True)
        pending.append((dist.all_reduce(slot_bufs[j], op=dist.ReduceOp.MIN, async_op=True), slot_bufs[j]))

    # --contexts M (single GPU): M independent contexts share the GPU, each with one tick in
    # flight -- what a host planning for several robots does.  A step still is one full tick.
    lps = [lp]
    inflight = [False]
    if args.contexts > 1:
        if world > 1 or scans is not None:
            raise SystemExit("--contexts needs --gpus 1 and a resident-cloud workload")
        for _ in range(args.contexts - 1):
            extra = LocalPlanner([theory], device=gpu, max_points=len(sc.cloud), max_trajectories=1 << 20)
            extra.set_cloud(sc.cloud)
            extra.setPlan(sc.plan)
            lps.append(extra)
            inflight.append(False)
    last_res = [None]
    overlap_marking = [marking is not None and args.marking_schedule == "overlapped"]

    def step():
        if len(lps) > 1:
            j = step_no[0] % len(lps)
            if inflight[j]:
                last_res[0] = lps[j].tick_end()
            lps[j].tick_begin(name, sc.tick)
            inflight[j] = True
            step_no[0] += 1
            return last_res[0]
        if scans is not None:
            scan = scans[step_no[0] % len(scans)]
            lp.set_scan(scan, t_bs, t_gb, 10.0, 2.0)
            if marking is not None:
                if overlap_marking[0]:
                    lp.tick_begin(name, sc.tick)
                    marking.update(scan, t_bs, t_gb)
                    res = lp.tick_end()
                    step_no[0] += 1
                    return res
                marking.update(scan, t_bs, t_gb)
        if reduce_mode == "inlib":
            res = lp.tick(name, sc.tick)                     # k_score -> ncclAllReduce -> resolve kernel, one stream
        elif reduce_mode == "torch":
            lp.tick_begin(name, sc.tick)                     # GPU computes tick i ...
            collect()                                        # ... while the host finishes tick i-2's all-reduce
            issue()                                          # ... and starts tick i-1's
            res = lp.tick_end()
            prev_words[0] = lp.winner_words(res)
            res = resolved[0] if resolved[0] is not None else res
        else:
            res = lp.tick(name, sc.tick)
        step_no[0] += 1
        return res

    def fence():
        for j in range(len(lps)):
            if inflight[j]:
                last_res[0] = lps[j].tick_end()
                inflight[j] = False
        if reduce_mode == "torch":
            collect()
            issue()
            collect()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    fence()
    dev_ms, score_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        lr = lp.last_result
        if lr is not None and lr.score_ms > 0:
            score_ms.append(lr.score_ms)              # latest sampled HIP-event duration of k_score
            dev_ms.append(lr.device_ms)               # ... and of the whole tick's kernels
    fence()
    elapsed = time.perf_counter() - t0
    serial_ms, marking_summary = None, None
    if marking is not None:
        # the other schedule, and the update's HIP-event times with nothing else on the GPU (right away: the GPU is warm)
        was = overlap_marking[0]
        overlap_marking[0] = False
        for _ in range(10):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(100):
            res = step()
        fence()
        serial_ms = (time.perf_counter() - t1) / 100 * 1e3
        overlap_marking[0] = was
        marking_summary = marking.summary()
        # (the parity check below looks at the cloud and the result of the LAST step, whichever schedule ran it)
    if reduce_mode == "torch":
        res = resolved[0]
    if len(lps) > 1:
        res = last_res[0]
    if world > 1:
        et = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())

    n_global = int(lp.last_result.n_samples)
    n_local = int(lp.last_result.n_local)
    value = n_global * args.steps / elapsed

    # ---- synchronous latency of one tick incl. the exchange (what a control loop can use) ----
    sync_ms = None
    if world > 1:
        lat = []
        for _ in range(60):
            dist.barrier()
            t1 = time.perf_counter()
            r1 = lp.tick(name, sc.tick)
            if reduce_mode == "torch":
                h = slot_host[0]
                h.fill_(sharding.INT64_MAX)
                w = lp.winner_words(r1)
                h[2 * rank], h[2 * rank + 1] = w
                slot_bufs[0].copy_(h)
                dist.all_reduce(slot_bufs[0], op=dist.ReduceOp.MIN)
                r1 = lp.resolve_words(slot_bufs[0].tolist())
            lat.append((time.perf_counter() - t1) * 1e3)
        lt = torch.tensor([median(lat[10:])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(lt, op=dist.ReduceOp.MAX)
        sync_ms = float(lt.item())
        res = r1

    ceiling = None
    if rank == 0 and not args.no_ceiling:
        cp, rd = lp.stream_ceiling(1 << 30, 10)
        ceiling = {"copy_GBps": round(cp, 1), "read_GBps": round(rd, 1), "buffer_bytes": 1 << 30,
                   "note": "float4 grid-stride kernels of this library on this GPU (copy counts read + write bytes)"}

    # ---- what a control loop sees (one GPU, resident-cloud workloads): inputs that move, the host boundary, the shipped sizes ----
    extras = {}
    if rank == 0 and world == 1 and scans is None and len(lps) == 1 and not args.no_extras:
        import oracle as _oracle
        # (a) moving inputs: pose / twist advance along the prune plan every tick, the cloud is replaced every k ticks
        #     (set_cloud of the other variant, H2D included); the last tick is checked against the oracle
        n_mv = max(60, min(args.steps, 200))
        mv_ticks, mv_clouds = moving_inputs(scenes, sc, n_mv + 10)
        for k in range(10):
            lp.tick(name, mv_ticks[k])
        torch.cuda.synchronize()
        t_ticks, n_sum, which = 0.0, 0, 0
        t0 = time.perf_counter()
        for k in range(10, n_mv + 10):
            if (k - 10) % args.replace_every == 0 and k > 10:
                which ^= 1
                lp.set_cloud(mv_clouds[which])
            t1 = time.perf_counter()
            mv_res = lp.tick(name, mv_ticks[k])
            t_ticks += time.perf_counter() - t1
            n_sum += int(mv_res.n_samples)
        t_mv = time.perf_counter() - t0
        mo = _oracle.tick(theory, mv_clouds[which], sc.plan, mv_ticks[n_mv + 9], n_threads=os.cpu_count() or 1).result
        mv_ok = bool(mv_res.planner_state == mo.planner_state and int(mv_res.best_index) == int(mo.best_index) and
                     abs(mv_res.vx - mo.vx) <= 1e-4 and abs(mv_res.vy - mo.vy) <= 1e-4 and abs(mv_res.wz - mo.wz) <= 1e-4 and
                     (mv_res.best_index < 0 or abs(mv_res.best_cost - mo.best_cost) <= 1e-4))
        extras["moving"] = {
            "value": round(n_sum / t_mv, 1), "ticks_only_value": round(n_sum / t_ticks, 1), "unit": "trajectories/s",
            "ticks": n_mv, "ms_per_tick": round(t_mv / n_mv * 1e3, 5), "ms_per_tick_ticks_only": round(t_ticks / n_mv * 1e3, 5),
            "mean_trajectories_per_tick": round(n_sum / n_mv, 1), "cloud_replaced_every": args.replace_every,
            "last_tick_matches_oracle": mv_ok, "last_tick_best_index": int(mv_res.best_index), "oracle_best_index": int(mo.best_index),
            "note": "pose / twist advance along the prune plan every tick; set_cloud (H2D of the other cloud variant) every "
                    f"{args.replace_every} ticks is inside `value`, outside `ticks_only_value`"}
        lp.set_cloud(sc.cloud)
        # (b) end to end through the host boundary: the reference re-aggregates the cloud every tick
        #     (local_planner.cpp:498-511), so: set_cloud from pageable 32-byte pcl::PointXYZI records + tick + result
        xyzi32 = np.zeros((len(sc.cloud), 8), np.float32)
        xyzi32[:, :4] = sc.cloud
        for _ in range(3):
            lp.set_cloud(xyzi32); lp.tick(name, sc.tick)
        n_e2e = 30
        t_set = 0.0
        t0 = time.perf_counter()
        for _ in range(n_e2e):
            t1 = time.perf_counter()
            lp.set_cloud(xyzi32)
            t_set += time.perf_counter() - t1
            e_res = lp.tick(name, sc.tick)
        t_e2e = (time.perf_counter() - t0) / n_e2e
        extras["end_to_end"] = {
            "ms": round(t_e2e * 1e3, 5), "value": round(int(e_res.n_samples) / t_e2e, 1), "unit": "trajectories/s",
            "set_cloud_ms": round(t_set / n_e2e * 1e3, 5), "cloud_bytes": int(xyzi32.nbytes),
            "what": "set_cloud from pageable host memory, 32-byte pcl::PointXYZI records (repack + H2D) + tick + result on the host, every tick"}
        lp.set_cloud(sc.cloud)
        lp.tick(name, sc.tick)
        # (c) the sizes the reference ships (55 / 275 / 2 x 126), warm and after 100 ms of idling
        extras["shipped_tick_latency"] = shipped_latencies(np, scenes, configs, LocalPlanner, gpu)

    rank_info = None
    if world > 1:
        pr = torch.cuda.get_device_properties(gpu)
        mine = {"rank": rank, "device": gpu, "name": pr.name,
                "pci": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', -1):02x}:{getattr(pr, 'pci_device_id', -1):02x}",
                "uuid": str(getattr(pr, "uuid", ""))}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rank_info = gathered
        if rank == 0 and args.backend == "nccl" and len({g["pci"] for g in gathered}) != world:
            print(f"[bench] WARNING: {world} ranks on fewer distinct GPUs: {[g['pci'] for g in gathered]}", file=sys.stderr, flush=True)

    out = None
    if rank == 0:
        import oracle
        ncpu = os.cpu_count() or 1
        ocloud = sc.cloud
        if scans is not None:       # the cloud the last tick actually scored
            ocloud = np.ascontiguousarray(lp.get_cloud())
        # ---- parity spot check of what was just timed: the oracle on the WHOLE batch ----
        ofull = oracle.tick(theory, ocloud, sc.plan, sc.tick, n_threads=ncpu)
        rf = ofull.result
        same_index = int(res.best_index) == int(rf.best_index)
        near_tie = (not same_index and res.best_index >= 0 and rf.best_index >= 0 and
                    abs(res.best_cost - rf.best_cost) <= 1e-6)
        # strict: the same winner AND its command / cost within 1e-4 (a near-tie that picked another index is reported
        # by winner_near_tie_within_1e-6 alone, never as a match)
        parity_ok = bool(res.planner_state == rf.planner_state and same_index and
                         abs(res.vx - rf.vx) <= 1e-4 and abs(res.vy - rf.vy) <= 1e-4 and abs(res.wz - rf.wz) <= 1e-4 and
                         (res.best_index < 0 or abs(res.best_cost - rf.best_cost) <= 1e-4))
        # ---- roofline of the dominant kernel (k_score) ----
        # (a) HBM roofline with the bytes THIS kernel must move once (compulsory traffic): the
        #     rollout state it reads (24 B per trajectory-step), the cell-sorted tile points (12 B),
        #     the row-run index, headers in / costs, steps, samples, load out (32 + 32 B per trajectory).
        # (b) SURVEY 8d's reference-equivalent gather bytes (32 per evaluated step + 16 per
        #     radius-search neighbour the reference materialises), which this design never moves:
        #     reported as a work rate, NOT as a fraction of HBM (it exceeded 1 in round 1).
        b0, e0 = sharding.shard_range(0, world, n_global)
        o = ofull if world == 1 else oracle.tick(theory, ocloud, sc.plan, sc.tick, begin=b0, end=e0, n_threads=ncpu)
        r = o.result
        units = int(r.steps_total)                         # trajectory-steps per launch (rank 0's shard)
        n_tile = int(lp.last_result.n_points_binned)
        compulsory = 24 * units + 12 * n_tile + 4 * 4096 + 64 * n_local + 16 * len(sc.plan)
        ref_bytes = 32 * int(r.steps_eval) + 16 * int(r.k_sum) + 16 * len(ocloud) + 32 * len(sc.plan) + 32 * n_local   # SURVEY 8d B_alg, 16 P included
        k_ms = float(np.mean(score_ms)) if score_ms else float("nan")
        t_ms = float(np.mean(dev_ms)) if dev_ms else float("nan")
        achieved = compulsory / (k_ms * 1e-3) / 1e9
        prof = {}
        prof_src = {}
        for fn, cands in (("traffic.json", ("traffic.json",)), ("pmc.json", ("r03_pmc.json", "r02_pmc.json")),
                          ("kernel_avg.json", ("r03_kernel_avg.json", "r02_kernel_avg.json"))):
            prof[fn] = {}
            for cand in cands:                                # the newest round's table that has this workload
                pth = os.path.join(ROOT, "profiles", cand)
                if not os.path.exists(pth):
                    continue
                try:
                    got = json.load(open(pth)).get(f"{workload}" if layout == scenes.DEFAULT_LAYOUT else f"{workload}_{layout}", {})
                except Exception:
                    got = {}
                if got:
                    prof[fn], prof_src[fn] = got, "profiles/" + cand
                    break
        traffic = prof.get("traffic.json", {}).get("k_score_hbm_bytes_per_launch") if world == 1 else None
        pmc = prof.get("pmc.json", {})
        roofline = {
            "bound": "hbm", "kernel": "k_score",
            "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
            "traffic": traffic,
            "traffic_static": {"static": True, "source": "profiles/traffic.json (PMC passes of an earlier run of this workload, not this run)"},
            # SURVEY 8d's own fraction: B_alg (what the reference's radius searches gather: 32 B per evaluated step + 16 B per
            # neighbour + 16 P + 32 M + 32 N) over the tick / the kernel, / 8 TB/s.  A work rate against the HBM peak: this
            # design never moves those bytes (frac above is the physical one).
            "frac_alg_tick": round(ref_bytes / (elapsed / args.steps) / 8.0e12, 4),
            "frac_alg_kernel": round(ref_bytes / (k_ms * 1e-3) / 8.0e12, 4) if k_ms == k_ms else None,
            "b_alg_bytes": ref_bytes,
            "bytes_per_launch": compulsory, "units_per_launch": units,
            "bytes_per_unit": round(compulsory / max(units, 1), 2),
            "kernel_ms": round(k_ms, 5), "kernel_ms_rocprofv3": prof.get("kernel_avg.json", {}).get("k_score_ms"),
            "kernel_ms_rocprofv3_static": True, "kernel_ms_rocprofv3_source": prof_src.get("kernel_avg.json"),
            "tick_device_ms": round(t_ms, 5),
            "what_binds_it": {
                "name": "VALU issue + memory/LDS latency (not HBM: the working set is L2/MALL resident)",
                "valu_busy_frac_of_issue_cycles": pmc.get("valu_busy_frac"),
                "waves_waiting_frac": pmc.get("waves_waiting_frac"),
                "static": True, "source": (prof_src.get("pmc.json", "") + " (an earlier run, not this one)") if pmc else None},
            "stream_ceiling": ceiling,
            "frac_of_copy_ceiling": round(achieved / ceiling["copy_GBps"], 4) if ceiling else None,
            "reference_equivalent_gather": {
                "bytes_per_launch": ref_bytes, "bytes_per_unit": round(ref_bytes / max(units, 1), 2),
                "GBps": round(ref_bytes / (k_ms * 1e-3) / 1e9, 1),
                "note": "SURVEY 8d B_alg (k_sum radius-search neighbours x 16 B + 32 B per evaluated step): what the "
                        "reference's kd-tree gathers would stream; the grid walk never materialises them, so this "
                        "is a work rate and may exceed the HBM peak"},
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline and scans is None:
            # the oracle ("port"), ONE core like the reference's loops A and B
            # (local_planner.cpp:549-557, 456-469); median tick of a bounded sample
            for _ in range(1 if base in ("C3", "C3P", "C4") else 3):
                oracle.tick(theory, sc.cloud, sc.plan, sc.tick, n_threads=1)
            ticks, kd, gen, scr, t_cpu = [], [], [], [], 0.0
            while (t_cpu < args.cpu_seconds and len(ticks) < 20) or len(ticks) < 3:
                t1 = time.perf_counter()
                oc = oracle.tick(theory, sc.cloud, sc.plan, sc.tick, n_threads=1)
                dt = time.perf_counter() - t1
                t_cpu += dt
                ticks.append(dt)
                kd.append(oc.result.t_kdtree_s)
                gen.append(oc.result.t_generate_s)
                scr.append(oc.result.t_score_s)
            alls = []
            for _ in range(3):
                t1 = time.perf_counter()
                oracle.tick(theory, sc.cloud, sc.plan, sc.tick, n_threads=ncpu)
                alls.append(time.perf_counter() - t1)
            cpu = {"value": round(n_global / median(ticks), 1), "unit": "trajectories/s", "cores": 1, "kind": "port",
                   "sample": f"median of {len(ticks)} full {workload} ticks ({n_global} traj x {n_steps_traj} steps) after warm-up, "
                             f"{t_cpu:.1f} s of CPU; oracle built -O3 -march=x86-64-v3 (the reference's Docker build passes no -march)",
                   "tick_ms": round(median(ticks) * 1e3, 2), "kdtree_build_ms": round(median(kd) * 1e3, 2),
                   "rollout_ms": round(median(gen) * 1e3, 2), "scoring_ms": round(median(scr) * 1e3, 2),
                   "all_cores_value": round(n_global / median(alls), 1), "all_cores": ncpu,
                   "all_cores_note": "rollout + scoring threaded over trajectories, kd-tree build serial"}
        if world == 1 and not args.no_cpu_baseline and scans is not None:
            # C5 / C5M: the oracle's whole step on ONE core -- cbSensor feed (crop + VoxelGrid) + [selfClear / selfMark of the
            # marking layer] + the C2 tick on the resulting cloud -- over the same scans, a bounded sample
            mo = None
            if marking is not None:
                from dddmr_navigation_amd import marking as marking_mod
                walls = sc.cloud[(np.abs(np.abs(sc.cloud[:, 1]) - 9.9) < 0.05)]
                mo = oracle.MarkingOracle(marking_mod.shipped_config(perception_window_size=10.0), marking_mod.ground_lattice(), walls[:, :3])
            steps_cpu, t_feed, t_mark, t_tick, t_cpu, i = [], [], [], [], 0.0, 0
            while (t_cpu < args.cpu_seconds and len(steps_cpu) < 20) or len(steps_cpu) < 3:
                scan = scans[i % len(scans)]
                i += 1
                t1 = time.perf_counter()
                obs_cpu = oracle.feed(scan[:, :3], t_bs, t_gb, 10.0, 2.0)
                t2 = time.perf_counter()
                if mo is not None:
                    mo.update(obs_cpu[:, :3], t_bs, t_gb)
                t3 = time.perf_counter()
                oracle.tick(theory, obs_cpu, sc.plan, sc.tick, n_threads=1)
                t4 = time.perf_counter()
                if i > 1:                      # (the first step warms caches and fills the marking store)
                    steps_cpu.append(t4 - t1); t_feed.append(t2 - t1); t_mark.append(t3 - t2); t_tick.append(t4 - t3)
                t_cpu += t4 - t1
            cpu = {"value": round(n_global / median(steps_cpu), 1), "unit": "trajectories/s", "cores": 1, "kind": "port",
                   "sample": f"median of {len(steps_cpu)} {workload} steps (feed of a 16x1800 scan"
                             + (" + marking / clearing update" if mo is not None else "") + f" + C2 tick of {n_global} trajectories) "
                             f"after one warm-up step, {t_cpu:.1f} s of CPU; oracle built -O3 -march=x86-64-v3",
                   "step_ms": round(median(steps_cpu) * 1e3, 2), "feed_ms": round(median(t_feed) * 1e3, 2),
                   "marking_update_ms": round(median(t_mark) * 1e3, 2) if mo is not None else None,
                   "tick_ms": round(median(t_tick) * 1e3, 2)}
        colliding = float((ofull.costs == -1.0).mean())
        out = {
            "metric": "scored trajectories/sec (N_traj x N_steps)", "value": round(value, 1),
            "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}: {n_global} trajectories x {n_steps_traj} steps vs "
                                   f"{len(ocloud)}-point cloud, {len(sc.plan)}-pose prune plan, shipped critic stack"
                                   + (", 16x1800 LiDAR scan -> set_scan (voxel-hash feed) fused into every step" if scans is not None else "")
                                   + (" + global-mode mark/clear layer on the same scan" if marking is not None else ""),
                       "scene_layout": layout, "colliding_share": round(colliding, 4),
                       "trajectories_per_gpu": n_local, "steps_per_trajectory": n_steps_traj,
                       "trajectory_steps_per_s": round(value * n_steps_traj, 1),
                       "parallelism": f"traj-shard x{world}" if world > 1 else ("single" if len(lps) == 1 else f"{len(lps)} independent contexts on one GPU, one tick in flight each"),
                       "contexts_per_gpu": len(lps),
                       "key_reduce": None if world == 1 else (
                           ("in-library RCCL ncclAllReduce(min) + resolve kernel on the tick's stream" if reduce_mode == "inlib"
                            else ("RCCL" if args.backend == "nccl" else args.backend) + " all_reduce(MIN) via torch.distributed, pipelined two ticks deep")
                           + f", {16 * world} bytes per tick (cost bits, -index per rank)"),
                       "sync_tick_latency_ms": None if sync_ms is None else round(sync_ms, 5),
                       "cmd_vel": [res.vx, res.vy, res.wz], "best_index": int(res.best_index),
                       "oracle_best_index": int(rf.best_index),
                       "cmd_vel_matches_oracle": parity_ok, "winner_near_tie_within_1e-6": bool(near_tie)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if marking is not None:
            ms = marking_summary
            ms["schedule"] = args.marking_schedule
            ms["serial_schedule_ms_per_step"] = round(serial_ms, 5)
            ms["note"] = ("clear_ms / mark_ms: HIP events of an update running alone (serial schedule); `value` is measured on the "
                          "schedule named here")
            out["config"]["marking"] = ms
            # the marking / clearing update against the HBM roofline: bytes it must move once (observation read by the grid
            # count, the scatter, the union-find and the partitions, 16 B each; grid cells; the store's slots; generator
            # points out and in; the window's ground nodes with their dGraph / lethal entries) over its HIP-event time.
            n_obs = len(ocloud)
            upd_ms = ms["clear_ms"] + ms["mark_ms"]
            b_upd = (4 * 16 + 8 + 4) * n_obs + 2 * 4 * 65536 + 32 * ms["alive_markings"] + 2 * 32 * int(ms["marked_per_update"] * 1.3) + 25 * 8000
            out["roofline_marking"] = {
                "bound": "hbm", "kernel": "marking update (5 launches: k_mkf_*)", "achieved": round(b_upd / (upd_ms * 1e-3) / 1e9, 2) if upd_ms > 0 else None,
                "peak": 8000.0, "unit": "GB/s", "frac": round(b_upd / (upd_ms * 1e-3) / 8.0e12, 5) if upd_ms > 0 else None,
                "bytes_per_update": b_upd, "update_ms": round(upd_ms, 5), "traffic": None,
                "what_binds_it": {"name": "dependent-latency chains and instruction issue (ray marches, union-find, 64 one-workgroup partitions), "
                                          "not HBM: the update's working set is a few MB", "static": True,
                                  "source": "profiles/r03_C5M_fused_pmc.json, profiles/r03_C5M_fused_kernel_stats.csv"}}
        out.update(extras)
        if "moving" in extras:
            out["config"]["moving_value"] = extras["moving"]["value"]
            if args.inputs == "moving":
                out["config"]["static_value"] = out["value"]
                out["value"] = extras["moving"]["value"]
                out["ms_per_step"] = extras["moving"]["ms_per_tick"]
                out["config"]["inputs"] = "moving (value = the moving-input rate; static_value = identical ticks)"
        if world > 1:
            out["config"]["ranks_seen"] = {"torch_world_size": dist.get_world_size(), "backend": args.backend, "ranks": rank_info,
                                           "distinct_gpus": len({g["pci"] for g in rank_info}),
                                           "inlib_communicator_ranks": lp.comm_ranks() if reduce_mode == "inlib" else None}

    # ---- the in-library RCCL exchange, measured after the timed loop (`--reduce auto`, RCCL backend) ----
    # k_score -> ncclAllReduce(min) -> k_resolve on the tick's stream: the synchronous tick a C++ host gets without
    # any collective code of its own.  A watchdog prints the line gathered so far and leaves if the bootstrap or a
    # collective never returns (first use of this path on a multi-GPU node).
    if world > 1 and reduce_mode == "torch" and args.reduce == "auto" and args.backend == "nccl":
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(90.0):
                # a collective that never returns: the line gathered so far is printed (marked failed), then EVERY rank
                # leaves with a non-zero code -- a hung GPU exchange is a failure of the run, not a footnote
                print(f"[bench] rank {rank}: in-library RCCL exchange did not return within 90 s", file=sys.stderr, flush=True)
                if out is not None:
                    out["config"]["inlib_rccl"] = {"status": "FAILED: timed out after 90 s"}
                    out["failed"] = "in-library RCCL exchange hung"
                    json_out.write(json.dumps(out) + "\n")
                    json_out.flush()
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        inlib = {"status": "communicator did not come up on every rank"}
        try:
            dist.barrier()
            if inlib_bootstrap():
                lat, r2 = [], None
                for _ in range(110):
                    dist.barrier()
                    t1 = time.perf_counter()
                    r2 = lp.tick(name, sc.tick)          # returns the GLOBAL winner on every rank
                    lat.append((time.perf_counter() - t1) * 1e3)
                lt = torch.tensor([median(lat[10:])], dtype=torch.float64, device=red_dev)
                dist.all_reduce(lt, op=dist.ReduceOp.MAX)
                inlib = {"status": "ok", "communicator_ranks": lp.comm_ranks(), "sync_tick_ms": round(float(lt.item()), 5),
                         "trajectories_per_s_synchronous": round(n_global / (float(lt.item()) * 1e-3), 1),
                         "winner_equals_torch_path": bool(r2.best_index == res.best_index and r2.best_cost == res.best_cost),
                         "note": "k_score -> ncclAllReduce(ncclInt64, ncclMin) of the slot vector -> k_resolve on the context's stream"}
                lp.comm_destroy()
        except Exception as e:      # noqa: BLE001 -- report, never lose the line
            inlib = {"status": f"{type(e).__name__}: {e}"}
        done.set()
        if out is not None:
            out["config"]["inlib_rccl"] = inlib
    for extra in lps[1:]:
        extra.close()
    if marking is not None:
        marking.close()
    lp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()


if __name__ == "__main__":
    main()
