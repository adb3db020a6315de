"""Ad-hoc GPU-vs-oracle comparison used while developing (tests/ hold the real
parity tests).  Usage: python tools/parity_check.py [playground C1 C2 C3 ...]"""
import sys, time
import numpy as np
from dddmr_navigation_amd import scenes, configs, _capi as K
from dddmr_navigation_amd.local_planner import LocalPlanner
import oracle


def compare(sc, name, n_threads=8, verbose=True):
    lp = LocalPlanner([sc.theory], max_points=max(len(sc.cloud), 16))
    lp.set_cloud(sc.cloud)
    lp.setPlan(sc.plan)
    tname = sc.theory.name.decode()
    res = lp.tick(tname, sc.tick)
    costs, steps, smp = lp.debug()
    t0 = time.time()
    o = oracle.tick(sc.theory, sc.cloud, sc.plan, sc.tick, n_threads=n_threads, want_margin=True)
    t_or = time.time() - t0
    r = o.result
    ok = True
    if not np.array_equal(steps, o.steps):
        bad = np.nonzero(steps != o.steps)[0]
        print(f"[{name}] STEP MISMATCH at {bad[:10]}: gpu {steps[bad[:10]]} oracle {o.steps[bad[:10]]}")
        ok = False
    if not np.array_equal(smp, o.samples):
        print(f"[{name}] SAMPLE MISMATCH"); ok = False
    neg_g, neg_o = costs < 0, o.costs < 0
    fragile = np.abs(o.min_margin) < 1e-4
    code_mis = np.nonzero((costs != o.costs) & (neg_g | neg_o))[0]
    hard = [i for i in code_mis if not fragile[i]]
    if len(hard):
        print(f"[{name}] REJECT-CODE MISMATCH (non-fragile) at {hard[:10]}: gpu {costs[hard[:10]]} oracle {o.costs[hard[:10]]} margin {o.min_margin[hard[:10]]}")
        ok = False
    both = ~neg_g & ~neg_o
    dmax = float(np.max(np.abs(costs[both] - o.costs[both]))) if both.any() else 0.0
    if dmax > 1e-4:
        i = int(np.argmax(np.abs(np.where(both, costs - o.costs, 0))))
        print(f"[{name}] COST MISMATCH max {dmax:.3e} at {i}: gpu {costs[i]} oracle {o.costs[i]}"); ok = False
    cmd_ok = (res.best_index == r.best_index and abs(res.vx - r.vx) <= 1e-4 and abs(res.vy - r.vy) <= 1e-4
              and abs(res.wz - r.wz) <= 1e-4 and res.planner_state == r.planner_state)
    if not cmd_ok:
        print(f"[{name}] CMD MISMATCH gpu idx {res.best_index} ({res.vx},{res.vy},{res.wz}) cost {res.best_cost} | oracle idx {r.best_index} ({r.vx},{r.vy},{r.wz}) cost {r.best_cost}")
        ok = False
    if verbose:
        print(f"[{name}] N={res.n_samples} binned={res.n_points_binned} device_ms={res.device_ms:.3f} "
              f"collide={neg_o.mean():.2f} fragile={int(fragile.sum())} code_mismatch={len(code_mis)} "
              f"max|dcost|={dmax:.2e} best={res.best_index} cost={res.best_cost:.6f} cmd=({res.vx:.4f},{res.vy:.4f},{res.wz:.4f}) "
              f"oracle_s={t_or:.2f} k_sum={r.k_sum} steps_eval={r.steps_eval} -> {'OK' if ok else 'FAIL'}")
    lp.close()
    return ok


def main():
    names = sys.argv[1:] or ["playground", "C1", "C2"]
    allok = True
    for n in names:
        if n == "playground":
            for goal in [(3.0, 1.0), (3.0, -1.0)]:
                for st in (5.0, 2.0):
                    allok &= compare(scenes.playground_scene(goal, st), f"playground{goal}{st}")
        else:
            allok &= compare(scenes.bench_scene(n), n)
    print("ALL OK" if allok else "SOME FAILED")
    sys.exit(0 if allok else 1)


if __name__ == "__main__":
    main()
