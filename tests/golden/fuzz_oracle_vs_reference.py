#!/usr/bin/env python3
"""Build-container only (needs /root/reference): random-shape campaign of the C oracle against the reference
environments themselves, beyond the committed suites.  Nothing is stored; every episode must agree bit for bit
(trace, rewards, f64 states, totals), exactly like tests/golden/make_golden.py checks its suites.

    python tests/golden/fuzz_oracle_vs_reference.py [--seconds 600] [--seed 0]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (sets up sys.path for the reference + the shim)

fi = G.fi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from environments.SO_FJSSP import SO_FJSSP_Environment
    from environments.MO_FJSSP_discretes import MO_FJSSP_Environment
    from environments.SO_SFJSP import SO_SFJSP_Environment
    from environments.MO_DFJSP_breakdown import MO_DFJSP_Environment
    from environments.SO_DFJSP import SO_DFJSP_Environment
    classes = {"so": SO_FJSSP_Environment, "mo": MO_FJSSP_Environment, "sf": SO_SFJSP_Environment, "dyn": MO_DFJSP_Environment,
               "sod": SO_DFJSP_Environment}
    flat = {"so": None, "mo": 18, "sf": 20, "dyn": "dyn", "sod": None}
    tmp = tempfile.mkdtemp(prefix="fjsp_fuzz_")
    rs = np.random.RandomState(args.seed)
    t_end = time.time() + args.seconds
    n_eps = n_steps = n_case = 0
    while time.time() < t_end:
        variant = ("so", "mo", "sf", "dyn", "sod")[n_case % 5]
        shape = ("small", "big", "jobs")[(n_case // 5) % 3]
        if shape == "jobs":
            R = int(rs.randint(1, 5)); Jlo = int(rs.randint(2, 5)); M = int(rs.randint(2, 9)); nmax = int(rs.randint(10, 41))
        elif shape == "big":
            R = int(rs.randint(8, 20)); Jlo = int(rs.randint(3, 7)); M = int(rs.randint(8, 33)); nmax = int(rs.randint(1, 3))
        else:
            R = int(rs.randint(1, 7)); Jlo = int(rs.randint(1, 4)); M = int(rs.randint(1, 13)); nmax = int(rs.randint(1, 5))
        S = int(rs.randint(1, 4)) if variant in ("so", "dyn", "sod") else 1
        prm = fi.GenParams(R_min=R, R_max=R, J_min=Jlo, J_max=Jlo + int(rs.randint(0, 2)), M=M, p_min=1, p_max=int(rs.randint(2, 60)),
                           N_min=1, N_max=nmax, S=S, DDT=float(rs.choice([0.5, 1.0, 1.5])), t_si_min=20.0, t_si_max=80.0)
        seed = int(rs.randint(1, 1 << 30))
        g = fi.InstanceSet(1).generate(0, seed, prm)
        while variant in ("dyn", "sod") and not (g.arrays(0).p > 0).any(axis=0).all():
            seed += 7919
            g.generate(0, seed, prm)
        if variant == "dyn":
            g.generate_machine_data(0, seed, max_windows=4, window_gap=(1, 60), window_len=(1, 30))
        folder = "F%06d" % n_case
        G.write_csv_folder(g.arrays(0), os.path.join(tmp, folder))
        s = fi.InstanceSet(1).load_csv(0, tmp, folder).solve_fluid()      # DDT parses like the reference's reader
        arr = s.arrays(0)
        Tmax = int((arr.count.sum(0) * arr.Jr).sum()) + 8
        for rep in range(2):
            actions = G.action_stream(("random",), int(rs.randint(1 << 30)), Tmax, flat[variant])
            rng_seed = int(rs.randint(1, 1 << 62))
            if variant == "so":
                mo = None
            elif variant in ("sf", "sod"):
                mo = variant
            elif variant == "mo":
                mo = (0.5, 0.5, 37.0, 91.0) if rep else (0.0, 1.0, None, None)
            else:
                mo = ("dyn", 3, 41.0, 17.0, 977.0) if rep else ("dyn", int(rs.randint(0, 3)), None, None, None)
            ref, env = G.run_reference(classes[variant], arr, tmp, folder, actions, rng_seed, check_lp=(n_eps % 8 == 0), mo=mo)
            if rep == 0:
                G.check_loader_against_reference(arr, env)
            ora = G.run_oracle(arr, actions, rng_seed, ref["T"], mo=mo)
            G.compare(ref, ora, "fuzz case %d %s/%s seed %d rep %d (R=%d M=%d S=%d K=%d)" % (n_case, variant, shape, seed, rep, arr.R, arr.M, arr.S, arr.K))
            n_eps += 1; n_steps += ref["T"]
        n_case += 1
        if n_case % 20 == 0:
            print("cases %d episodes %d steps %d  LP checks %d (max gap %.1e)" % (n_case, n_eps, n_steps, G.LP_STATS["solves"], G.LP_STATS["max_obj_gap"]), flush=True)
    print("DONE: %d cases, %d episodes, %d steps bit-exact; LP checks vs HiGHS %d, max objective gap %.2e, max infeasibility %.2e"
          % (n_case, n_eps, n_steps, G.LP_STATS["solves"], G.LP_STATS["max_obj_gap"], G.LP_STATS["max_infeas"]))


if __name__ == "__main__":
    main()
