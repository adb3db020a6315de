#!/usr/bin/env python3
"""bench.py -- scored trajectories/sec of the local-planner tick on MI355X.

A "step" is one control tick over one batch of synthetic input: local costmap
binning + fused rollout + all critics + argmin (+ one 8-byte min all-reduce when
N > 1), with the cloud already resident in HBM (set_cloud is outside the timed
region) and the chosen cmd_vel delivered to the host every tick.

Workload: BASELINE.json configs[1] ("C2": 4096 trajectories x 50 steps vs a
100k-point cloud) per GPU.  Multi-GPU is weak scaling: the global batch is
4096*N samples (x-axis of the sample grid 16*N long), rank r scores the
contiguous index range [4096 r, 4096 (r+1)), one all-reduce picks the winner.
`--workload C3|C4` selects the other configurations (C4 = 65536 samples, strong).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C2", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--backend", default=os.environ.get("DDDMR_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="independent planner contexts (robots) per GPU, one tick in flight each; 1 = sequential ticks (the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from dddmr_navigation_amd import scenes, configs, sharding, _capi as K
    from dddmr_navigation_amd.local_planner import LocalPlanner

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rollout engine has no CPU fallback")
    gpu = local_rank % torch.cuda.device_count()      # (several ranks may share a GPU only with --backend gloo)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = args.backend
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                # fail early if RCCL cannot move 8 bytes between the ranks
                probe = torch.full((1,), rank, dtype=torch.int64, device=dev)
                dist.all_reduce(probe, op=dist.ReduceOp.MIN)
                assert int(probe.item()) == 0
            except Exception as e:        # the 8-byte key reduce also works over gloo: say so and go on
                print(f"[bench] RCCL unavailable ({type(e).__name__}: {e}); falling back to gloo", file=sys.stderr, flush=True)
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
    base = "C2" if args.workload == "C5" else args.workload
    sc = scenes.bench_scene(base)
    theory = sc.theory
    scaling = "weak"
    if base == "C2":
        theory.linear_x_sample = 16.0 * world          # 4096 samples per GPU
    elif base == "C4":
        scaling = "strong"
    name = theory.name.decode()
    b = configs.BENCH[base]
    n_steps_traj = b["steps"]

    # HIP events serialise the queue around them (~3 us each): k_score is timed on every 8th tick
    os.environ.setdefault("DDDMR_TIMING_EVERY", "8")
    lp = LocalPlanner([theory], device=gpu, max_points=len(sc.cloud), max_trajectories=1 << 20,
                      rank=rank, world_size=world)
    lp.set_cloud(sc.cloud)            # inputs resident in HBM before the timed region
    lp.setPlan(sc.plan)
    scans = None
    if args.workload == "C5":
        # perception feed fused with the tick: 10 simulated 16-ring LiDAR scans of the C2
        # scene; every step = set_scan (crop + 0.1 m voxel-hash downsample on the GPU,
        # H2D of the raw scan included) + one C2 tick on the resulting cloud
        scans = [scenes.lidar_scan(sc.cloud, seed=100 + i) for i in range(10)]
        t_bs, t_gb = (0.0, 0.0, 0.5, 0, 0, 0, 1), (0.0, 0.0, 0.0, 0, 0, 0, 1)
    step_no = [0]

    # Multi-rank: everything the host does for the 8-byte min all-reduce -- collecting
    # the reduce of tick i-2, issuing the one of tick i-1 (RCCL over xGMI) -- happens
    # between tick_begin(i) and tick_end(i), i.e. while the GPU computes tick i, so the
    # collective costs the timed loop (almost) nothing.  The winner of a tick is
    # delivered two ticks later; every all-reduce completes inside the timed region
    # (fence() issues and collects the last ones).
    key_bufs = [torch.zeros(1, dtype=torch.int64, device=red_dev) for _ in range(2)]
    pending = []          # [(work handle, buffer)]
    resolved = [None]
    prev_key = [None]     # key of the last finished tick, not yet handed to the all-reduce
    n_issued = [0]

    def collect():
        while pending:
            work, buf = pending.pop(0)
            work.wait()
            resolved[0] = lp.resolve(int(buf.item()))

    def issue():
        if prev_key[0] is None:
            return
        buf = key_bufs[n_issued[0] % 2]
        n_issued[0] += 1
        buf.fill_(prev_key[0])
        prev_key[0] = None
        pending.append((dist.all_reduce(buf, op=dist.ReduceOp.MIN, async_op=True), buf))

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
            lp.set_scan(scans[step_no[0] % len(scans)], t_bs, t_gb, 10.0, 2.0)
        if world > 1:
            lp.tick_begin(name, sc.tick)                     # GPU computes tick i ...
            collect()                                        # ... while the host finishes tick i-2's all-reduce
            issue()                                          # ... and starts tick i-1's
            res = lp.tick_end()
            prev_key[0] = res.key
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
        if world > 1:
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
        if lr is not None:
            if lr.score_ms > 0:
                score_ms.append(lr.score_ms)              # latest sampled HIP-event duration of k_score
            dev_ms.append(lr.device_ms)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
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

    out = None
    if rank == 0:
        import oracle
        # ---- roofline of the dominant kernel (k_score), SURVEY.md 8(d) ----
        # algorithmic bytes of this rank's launch = sum over the steps the
        # reference evaluates of (32 + 16 k) + 32 M + 32 N_local, k = radius-search
        # result sizes, counted by the oracle on the very same inputs.
        b0, e0 = sharding.shard_range(0, world, n_global)
        ocloud = sc.cloud
        if scans is not None:       # the cloud the last tick actually scored
            g = lp.get_cloud()
            ocloud = np.ascontiguousarray(g)
        o = oracle.tick(theory, ocloud, sc.plan, sc.tick, begin=b0, end=e0, n_threads=os.cpu_count() or 1)
        r = o.result
        units = int(r.steps_total)                         # trajectory-steps per launch
        alg_bytes = 32 * int(r.steps_eval) + 16 * int(r.k_sum) + 32 * len(sc.plan) + 32 * n_local
        per_unit = alg_bytes / max(units, 1)
        k_ms = float(np.mean(score_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get("k_score_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_score", "achieved": round(achieved, 2), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "alg_bytes_per_launch": alg_bytes, "units_per_launch": units,
                    "bytes_per_unit": round(per_unit, 2), "kernel_ms": round(k_ms, 5),
                    "tick_device_ms": round(float(np.mean(dev_ms)), 5),
                    "tick_alg_bytes": alg_bytes + 16 * len(ocloud)}
        # parity spot check of what was just timed
        parity_ok = bool(world > 1 or scans is not None or (res.best_index == r.best_index and abs(res.vx - r.vx) <= 1e-4
                                       and abs(res.vy - r.vy) <= 1e-4 and abs(res.wz - r.wz) <= 1e-4))
        cpu = None
        if world == 1 and not args.no_cpu_baseline and scans is None:
            # the oracle ("port"), 1 core like the reference's loops, bounded sample
            n_ticks, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds:
                t1 = time.perf_counter()
                oracle.tick(theory, sc.cloud, sc.plan, sc.tick, n_threads=1)
                t_cpu += time.perf_counter() - t1
                n_ticks += 1
            ncpu = os.cpu_count() or 1
            t1 = time.perf_counter()
            oracle.tick(theory, sc.cloud, sc.plan, sc.tick, n_threads=ncpu)
            t_all = time.perf_counter() - t1
            cpu = {"value": round(n_global * n_ticks / t_cpu, 1), "unit": "trajectories/s", "cores": 1,
                   "kind": "port", "sample": f"{n_ticks} full {args.workload} ticks ({n_global} traj x {n_steps_traj} steps, "
                   f"kd-tree build included), {t_cpu:.1f} s", "all_cores_value": round(n_global / t_all, 1),
                   "all_cores": ncpu}
        out = {
            "metric": "scored trajectories/sec (N_traj x N_steps)", "value": round(value, 1),
            "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n_global} trajectories x {n_steps_traj} steps vs "
                                   f"{len(ocloud)}-point cloud, {len(sc.plan)}-pose prune plan, shipped critic stack"
                                   + (", 16x1800 LiDAR scan -> set_scan (voxel-hash feed) fused into every step" if scans is not None else ""),
                       "trajectories_per_gpu": n_local, "steps_per_trajectory": n_steps_traj,
                       "trajectory_steps_per_s": round(value * n_steps_traj, 1),
                       "parallelism": f"traj-shard x{world}" if world > 1 else ("single" if len(lps) == 1 else f"{len(lps)} independent contexts on one GPU, one tick in flight each"),
                       "contexts_per_gpu": len(lps),
                       "key_reduce": (("RCCL" if args.backend == "nccl" else args.backend) + " all_reduce(MIN), 8 bytes per tick") if world > 1 else None,
                       "cmd_vel": [res.vx, res.vy, res.wz], "best_index": int(res.best_index),
                       "cmd_vel_matches_oracle": parity_ok},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    for extra in lps[1:]:
        extra.close()
    lp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
