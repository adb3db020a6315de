"""One rank of tests/test_comm_gpu.py::test_rccl_ranks_on_separate_gpus (a child process per GPU).
usage: python comm_rank.py <scene> <rank> <world> <dir>
Rank 0 writes the communicator's unique id to <dir>/id.bin; every rank creates its shard's context on GPU <rank>,
joins the communicator, ticks three times and leaves <dir>/rank<r>.json."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dddmr_navigation_amd import configs, scenes               # noqa: E402
from dddmr_navigation_amd.local_planner import LocalPlanner     # noqa: E402


def main():
    scene, rank, world, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sc = scenes.playground_scene() if scene == "playground" else scenes.bench_scene("C1")
    if scene == "rotate":
        sc.theory = configs.rotate_inplace_shipped("rot", shortest=True)
    name = sc.theory.name.decode()
    with LocalPlanner([sc.theory], device=rank, max_points=max(len(sc.cloud), 16), rank=rank, world_size=world) as lp:
        idf = os.path.join(d, "id.bin")
        if rank == 0:
            with open(idf + ".tmp", "wb") as f:
                f.write(lp.comm_unique_id())
            os.replace(idf + ".tmp", idf)
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id after 60 s")
            time.sleep(0.05)
        lp.set_cloud(sc.cloud)
        lp.setPlan(sc.plan)
        lp.comm_init(open(idf, "rb").read(), rank, world)
        out = {"comm_ranks": lp.comm_ranks(), "ticks": []}
        for _ in range(3):
            r = lp.tick(name, sc.tick)
            out["ticks"].append({"state": int(r.planner_state), "best_index": int(r.best_index), "best_cost": float(r.best_cost),
                                 "cmd": [float(r.vx), float(r.vy), float(r.wz)], "n_local": int(r.n_local), "n_samples": int(r.n_samples)})
        lp.comm_destroy()
    with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
