"""The map sweep driver — counterpart of swarm_planner/src/swarm_traj_planner_rbp_test_all.cpp:49-103.

For every map of the sweep: distance grid, ECBS initial trajectory (host front-end), then the two calls of the hot path
through the C ABI (Corridor::update, RBPPlanner::update), with the reference's per-stage log lines.  Two modes:

  --mode serial   one map at a time, the synchronous host-buffer calls (what the reference's main loop does)
  --mode batched  all maps resident in one device session, one launch per stage (what bench.py times)

usage: python -m swarm_simulator_amd.test_all [--mission mission_64agents_15.json] [--maps 1-50] [--mode batched]
       [--batch-size 4] [--iteration 1] [--joint] [--csv DIR]
"""
import argparse
import time

from . import host, planner
from .types import Param


def parse_maps(spec: str):
    out = []
    for part in spec.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mission", default="mission_64agents_15.json")
    ap.add_argument("--maps", default="1-50")
    ap.add_argument("--mode", choices=["serial", "batched"], default="serial")
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--iteration", type=int, default=1)
    ap.add_argument("--joint", action="store_true")
    ap.add_argument("--csv", default=None, help="write coef<qi>.csv per map into DIR/map<i>/ (generateCoefCSV)")
    args = ap.parse_args(argv)

    param = Param.test_sweep(batch_size=args.batch_size, iteration=args.iteration, sequential=not args.joint)
    mission = host.load_mission(args.mission)
    maps = parse_maps(args.maps)
    worlds, plans, ok_maps = [], [], []
    for i in maps:
        print(f"Map: map{i}.bt")
        t0 = time.perf_counter()
        w = host.load_world(f"map{i}.bt", param)
        print(f"Euclidean Distmap runtime: {time.perf_counter() - t0:.6f}")
        t0 = time.perf_counter()
        try:
            pr = host.ecbs_plan(w, mission, param)
        except RuntimeError as e:
            print(f"[ERROR] {e}")
            return -1
        print(f"Initial Trajectory Planner runtime: {time.perf_counter() - t0:.6f}")
        if args.mode == "serial":
            t0 = time.perf_counter()
            cor = planner.Corridor(w, mission, param)
            if not cor.update(param.log, pr):
                print(f"[ERROR] {cor.last_error}")
                return -1
            print(f"BoxGenerator runtime: {time.perf_counter() - t0:.6f}")
            t0 = time.perf_counter()
            pl = planner.RBPPlanner(mission, param)
            if not pl.update(param.log, pr):
                print(f"[ERROR] {pl.last_error}")
                return -1
            print(f"SwarmPlanner runtime: {time.perf_counter() - t0:.6f}")
            report(i, mission, param, pr, args.csv)
        worlds.append(w), plans.append(pr), ok_maps.append(i)
    if args.mode == "batched":
        # every map keeps its own M = makespan + 2 (ecbs_planner.hpp:41-43): the session is ragged, nothing is padded
        sess = planner.Session(worlds, [mission] * len(plans), param, plans)
        t0 = time.perf_counter()
        sess.run()
        status = sess.download()
        dt = time.perf_counter() - t0
        print(f"BoxGenerator + SwarmPlanner runtime, {len(plans)} maps in one session: {dt:.6f} "
              f"({len(plans) * mission.qn / dt:.1f} agent-trajectories/s incl. download)")
        for i, p, st in zip(ok_maps, plans, status):
            if st:
                print(f"[ERROR] map{i}: status {st}")
                return -1
            report(i, mission, param, p, args.csv)
        sess.close()
    return 0


def report(i, mission, param, plan, csv_dir):
    ratio, dist = host.validate(mission, param, plan)
    print(f"map{i}: QP total cost {plan.total_cost:.6f}  time_scale {plan.time_scale:.4f}  makespan {plan.T[-1]:.3f}  "
          f"safety margin ratio {ratio:.4f}  total flight distance {dist:.3f}")
    if csv_dir:
        import os
        host.write_coef_csv(os.path.join(csv_dir, f"map{i}"), plan)


if __name__ == "__main__":
    raise SystemExit(main())
