#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py            # regenerate + verify everything
    python tests/golden/make_golden.py --quick    # fewer verification-only cases

What it does, per case:
  1. builds the instance with the PRODUCT's host code (CSV loader or seeded
     generator + fluid LP: libfjsp_amd.so through the package's ctypes binding);
  2. imports the reference environment from /root/reference under
     oracle/ref_shim (openpyxl dummy; docplex stand-in whose solve() returns the
     product's x and whose recorded model is checked against scipy/HiGHS: x must
     be feasible for the model the reference built and attain its optimum);
  3. plays episodes on the reference (fresh env object per episode,
     random.choice replaced by the shared counter-based stream) and records the
     per-step trace;
  4. plays the same actions on the C oracle (oracle/fjsp_oracle.c) and demands
     BIT-EXACT equality of every recorded quantity (states as f64 bit patterns);
  5. writes a subset of the traces as fixtures (*.npz, numpy arrays only).

A fixture is data: instance arrays, x, actions, expected outputs.  No reference
source text is stored.
"""
import argparse
import csv
import hashlib
import os
import random
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
sys.path.insert(0, REF)

import matplotlib  # noqa: E402
matplotlib.use("Agg")

from deep_reinforcement_learning_for_fjsp_amd import instances as fi  # noqa: E402
from oracle import pyoracle  # noqa: E402
import docplex.mp.model as shim  # noqa: E402  (the stand-in)

MASK64 = (1 << 64) - 1


def splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


class ChoiceStream(object):
    """random.choice replacement shared with the oracle / kernels (fjsp_oracle.h)."""

    def __init__(self, seed):
        self.seed, self.calls = seed & MASK64, 0

    def __call__(self, seq):
        u = splitmix64((self.seed + self.calls) & MASK64)
        self.calls += 1
        return seq[((u >> 32) * len(seq)) >> 32]


# ---------------------------------------------------------------------------
def write_csv_folder(arr, folder):
    """Instance arrays -> the reference's CSV folder format (Instance_generate.py:96-119 layout,
    plus the DDT column Data.read needs, SO_DFJSP_instance_read.py:53)."""
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "based_data.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kind_count", "machine_count", "order_count", "DDT"])
        w.writerow([arr.R, arr.M, arr.S, arr.ddt])
    with open(os.path.join(folder, "process_data.csv"), "w", newline="") as f:
        w = csv.writer(f)
        dyn = hasattr(arr, "power")
        w.writerow(["kind", "task", "machine_selectable", "process_time"] + (["power"] if dyn else []))
        koff = arr.koff
        for r in range(arr.R):
            for j in range(int(arr.Jr[r])):
                k = int(koff[r]) + j
                ms = tuple(int(m) for m in arr.elig_list[k, :arr.elig_n[k]])
                ts = tuple(int(arr.p[k, m]) for m in ms)
                w.writerow([r, j, ms, ts] + ([tuple(int(arr.power[k, m]) for m in ms)] if dyn else []))
    if dyn:      # MO_DFJSP_instance_read.py:56-73: one row per breakdown window, or one bare row per machine
        with open(os.path.join(folder, "machine_data.csv"), "w", newline="") as f:
            w = csv.writer(f)
            off = np.concatenate(([0], np.cumsum(arr.bk_n)))
            w.writerow(["machine", "idle_power", "breakdown_start", "breakdown_end"] if off[-1] else ["machine", "idle_power"])
            for m in range(arr.M):
                if arr.bk_n[m] == 0:
                    w.writerow([m, int(arr.idle_power[m])])
                for q in range(int(off[m]), int(off[m + 1])):
                    w.writerow([m, int(arr.idle_power[m]), int(arr.bk[q, 0]), int(arr.bk[q, 1])])
    with open(os.path.join(folder, "order_data.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["order", "time_arrive", "time_delivery", "kind_number"])
        for s in range(arr.S):
            w.writerow([s, int(arr.arrive[s]), int(arr.delivery[s]), tuple(int(c) for c in arr.count[s])])


def check_loader_against_reference(arr, env):
    """The product loader must reproduce what Data.read/process produced."""
    assert env.kind_count == arr.R and env.machine_count == arr.M and env.order_count == arr.S
    assert env.kind_task_tuple == arr.kind_task_tuple
    koff = arr.koff
    for (r, j) in env.kind_task_tuple:
        k = int(koff[r]) + j
        assert tuple(env.machine_rj_dict[(r, j)]) == tuple(int(m) for m in arr.elig_list[k, :arr.elig_n[k]])
        for m in range(arr.M):
            ref = env.time_mrj_dict[m].get((r, j), 0)
            assert ref == int(arr.p[k, m]), (r, j, m)
    for s in range(arr.S):
        assert tuple(env.count_sr_dict[s]) == tuple(int(c) for c in arr.count[s])
        assert env.time_arrive_s_dict[s] == int(arr.arrive[s])
        assert env.time_delivery_s_dict[s] == int(arr.delivery[s])
    if hasattr(env, "breakdown_m_dict"):
        off = np.concatenate(([0], np.cumsum(arr.bk_n)))
        for m in range(arr.M):
            assert env.power_m_dict[m] == int(arr.idle_power[m])
            assert [tuple(w) for w in env.breakdown_m_dict[m]] == [tuple(int(v) for v in arr.bk[q]) for q in range(off[m], off[m + 1])]
            for (r, j), pw in env.power_mrj_dict[m].items():
                assert pw == int(arr.power[int(koff[r]) + j, m])


LP_STATS = {"solves": 0, "max_obj_gap": 0.0, "max_infeas": 0.0}


def make_solve_hook(arr, env_ref, check_lp=True):
    """SOLVE_HOOK for the docplex stand-in: product LP + optimality check vs HiGHS."""
    koff = arr.koff

    def hook(model):
        env = env_ref[0]
        Q = np.array([env.kind_task_dict[rj].fluid_unprocessed_number_start for rj in env.kind_task_tuple], np.int32)
        now = np.array([env.kind_task_dict[rj].fluid_number for rj in env.kind_task_tuple], np.int32)
        x, obj = fi.fluid_lp(arr.Jr, arr.p, Q, now)
        X = model.var_dicts[0]
        values = {key: float(x[int(koff[key[1][0]]) + key[1][1], key[0]]) for key in X}
        LP_STATS["solves"] += 1
        if check_lp:
            # the recorded model IS the reference's LP: check x against it
            infeas = 0.0
            for c in model.constraints:
                v = c.expr.const + sum(co * values[kk] for kk, co in c.expr.terms.items())
                infeas = max(infeas, v if c.sense == "<=" else -v)
            for key, var in X.items():
                infeas = max(infeas, var.lb - values[key], values[key] - var.ub)
            tval = min(e.const + sum(co * values[kk] for kk, co in e.terms.items()) for e in model.objective.exprs)
            from scipy.optimize import linprog
            keys = list(X)
            col = {kk: i for i, kk in enumerate(keys)}
            n = len(keys) + 1
            A, b = [], []
            for e in model.objective.exprs:          # t - expr <= 0
                row = np.zeros(n); row[-1] = 1.0
                for kk, co in e.terms.items():
                    row[col[kk]] -= co
                A.append(row); b.append(e.const)
            for c in model.constraints:
                row = np.zeros(n)
                for kk, co in c.expr.terms.items():
                    row[col[kk]] += co
                sgn = 1.0 if c.sense == "<=" else -1.0
                A.append(sgn * row); b.append(-sgn * c.expr.const)
            cvec = np.zeros(n); cvec[-1] = -1.0
            bounds = [(X[kk].lb, X[kk].ub) for kk in keys] + [(None, None)]
            res = linprog(cvec, A_ub=np.array(A), b_ub=np.array(b), bounds=bounds, method="highs-ds")
            assert res.status == 0, res.message
            gap = abs(-res.fun - tval) / max(1.0, abs(res.fun))
            LP_STATS["max_obj_gap"] = max(LP_STATS["max_obj_gap"], gap)
            LP_STATS["max_infeas"] = max(LP_STATS["max_infeas"], infeas)
            assert infeas < 1e-9, "product x infeasible for the reference's LP: %g" % infeas
            assert gap < 1e-8, "product x not optimal for the reference's LP: %g vs %g" % (tval, -res.fun)
        return values
    return hook


def action_stream(kind, seed, n, flat=None):
    if kind[0] == "fixed":
        return np.tile(np.array(kind[1], np.uint8), (n, 1))
    rs = np.random.RandomState(seed)
    if flat == "dyn":                          # MO_DFJSP_breakdown.py:32 actions_size = [12, 10]
        return np.stack([rs.randint(0, 12, n), rs.randint(0, 10, n)], 1).astype(np.uint8)
    if flat:                                   # flat action in column 0 (MO_FJSSP_discretes / SO_SFJSP)
        return np.stack([rs.randint(0, flat, n), np.zeros(n, np.int64)], 1).astype(np.uint8)
    return np.stack([rs.randint(0, 6, n), rs.randint(0, 5, n)], 1).astype(np.uint8)


def run_reference(EnvCls, arr, folder_parent, folder_name, actions, rng_seed, check_lp, timing=None, mo=None):
    env_ref = [None]
    shim.SOLVE_HOOK = make_solve_hook(arr, env_ref, check_lp)
    env = EnvCls(use_instance=False, path=folder_parent, file_name=folder_name)
    env_ref[0] = env
    stream = ChoiceStream(rng_seed)
    orig_choice = random.choice
    random.choice = stream
    rec = {"k": [], "m": [], "job_r": [], "job_n": [], "reward": [], "done": [], "step_time": [], "delay": [],
           "states": []}
    koff = arr.koff
    sel = {}
    orig_ts, orig_ms = env.task_select, env.machine_select

    def ts(rule):
        rj = orig_ts(rule)
        sel["rj"] = rj
        return rj

    def ms(rule, rj):
        m = orig_ms(rule, rj)
        sel["m"] = m
        job = env.kind_task_dict[rj].job_now_list[0]
        sel["job"] = (job.kind, job.number)
        return m
    env.task_select, env.machine_select = ts, ms
    try:
        state0 = np.array(env.reset(), dtype=np.float64)
        t = 0
        el = 0.0
        while not env.done:
            a = actions[t]
            t0 = time.perf_counter()
            if mo is None or mo == "sod":
                s, r, d = env.step([int(a[0]), int(a[1])])
            elif mo == "sf":
                s, r, d = env.step(int(a[0]))
            elif mo[0] == "dyn":
                s, r, d = env.step([int(a[0]), int(a[1])], reward_policy=mo[1], completion=mo[2], tardiness=mo[3],
                                   energy_consumption=mo[4])
            else:
                s, r, d = env.step(int(a[0]), weight_vector=(mo[0], mo[1]), completion=mo[2], tardiness=mo[3])
            el += time.perf_counter() - t0
            rec["k"].append(int(koff[sel["rj"][0]]) + sel["rj"][1]); rec["m"].append(sel["m"])
            rec["job_r"].append(sel["job"][0]); rec["job_n"].append(sel["job"][1])
            rec["reward"].append(float(r)); rec["done"].append(bool(d))
            rec["step_time"].append(env.step_time); rec["delay"].append(env.delay_time_sum)
            rec["states"].append(np.array(s, dtype=np.float64))
            t += 1
        if timing is not None:
            timing[0] += el; timing[1] += t
    finally:
        random.choice = orig_choice
    out = {kk: np.array(v) for kk, v in rec.items()}
    out["state0"] = state0
    out["tend"] = np.array([env.machine_dict[m].time_end for m in env.machine_tuple], np.int32)
    out["makespan"] = int(out["tend"].max())
    out["delay_time_sum"] = int(env.delay_time_sum)
    out["fluid_completed_time"] = float(getattr(env, "fluid_completed_time", -1.0))   # class_MODFJSP keeps none
    out["completion_time"] = int(getattr(env, "completion_time", 0))
    out["energy"] = int(getattr(env, "energy_consumption", 0))
    out["T"] = t
    return out, env


def run_oracle(arr, actions, rng_seed, T, mo=None):
    lp = lambda Q, now: fi.fluid_lp(arr.Jr, arr.p, Q, now)[0]
    if mo is None:
        env = pyoracle.OracleEnv(arr, lp, pyoracle.SO_FJSSP, rng_seed)
    elif mo == "sod":
        env = pyoracle.OracleEnv(arr, lp, pyoracle.SO_DFJSP, rng_seed)
    elif mo == "sf":
        env = pyoracle.OracleEnv(arr, lp, pyoracle.SO_SFJSP, rng_seed)
    elif mo[0] == "dyn":
        env = pyoracle.OracleEnv(arr, lp, pyoracle.MO_DFJSP, rng_seed, ddt=arr.ddt)
    else:
        env = pyoracle.OracleEnv(arr, lp, pyoracle.MO_FJSSP_DISCRETES, rng_seed, ddt=arr.ddt)
    rec = {"k": [], "m": [], "job_r": [], "job_n": [], "reward": [], "done": [], "step_time": [], "delay": [],
           "states": []}
    state0 = env.reset()
    t = 0
    while not env.done:
        if mo is None or mo == "sod":
            s, r, d = env.step(actions[t])
        elif mo == "sf":
            s, r, d = env.step_sf(int(actions[t][0]))
        elif mo[0] == "dyn":
            s, r, d = env.step_dyn(actions[t], mo[1], mo[2], mo[3], mo[4])
        else:
            s, r, d = env.step_mo(int(actions[t][0]), (mo[0], mo[1]), mo[2], mo[3])
        tr = env.trace
        rec["k"].append(tr.k_sel); rec["m"].append(tr.m_sel); rec["job_r"].append(tr.job_kind)
        rec["job_n"].append(tr.job_n); rec["reward"].append(r); rec["done"].append(d)
        rec["step_time"].append(tr.step_time); rec["delay"].append(tr.delay_time_sum); rec["states"].append(s)
        t += 1
        assert t <= T + 5, "oracle episode longer than the reference's"
    out = {kk: np.array(v) for kk, v in rec.items()}
    out["state0"] = state0
    out["tend"] = env.machine_time_end()
    out["makespan"] = env.makespan
    out["delay_time_sum"] = env.delay_time_sum
    out["fluid_completed_time"] = env.fluid_completed_time
    out["completion_time"] = env.completion_time
    out["energy"] = env.energy_consumption if mo is not None and mo not in ("sf", "sod") and mo[0] == "dyn" else 0
    out["T"] = t
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def compare(ref, ora, tag):
    assert ref["T"] == ora["T"], "%s: episode length %d vs %d" % (tag, ref["T"], ora["T"])
    for key in ("k", "m", "job_r", "job_n", "done", "step_time", "delay"):
        if not np.array_equal(ref[key].astype(np.int64), ora[key].astype(np.int64)):
            i = int(np.nonzero(ref[key].astype(np.int64) != ora[key].astype(np.int64))[0][0])
            raise AssertionError("%s: %s differs at step %d: ref %s oracle %s" % (tag, key, i, ref[key][i], ora[key][i]))
    assert np.array_equal(bits(ref["reward"]), bits(ora["reward"])), tag + ": reward"
    assert np.array_equal(bits(ref["state0"]), bits(ora["state0"])), tag + ": reset state"
    if ref["T"]:
        rb, ob = bits(ref["states"]), bits(ora["states"])
        if not np.array_equal(rb, ob):
            i, j = [int(v[0]) for v in np.nonzero(rb != ob)]
            raise AssertionError("%s: state[%d][%d] ref %r oracle %r" % (tag, i, j, ref["states"][i][j], ora["states"][i][j]))
    assert np.array_equal(ref["tend"], ora["tend"]), tag + ": machine time_end"
    assert ref["makespan"] == ora["makespan"] and ref["delay_time_sum"] == ora["delay_time_sum"], tag
    if ref["fluid_completed_time"] >= 0:
        assert ref["fluid_completed_time"] == ora["fluid_completed_time"], tag + ": fluid_completed_time"
    if ref.get("completion_time"):
        assert ref["completion_time"] == ora["completion_time"], tag + ": completion_time"
    assert ref.get("energy", 0) == ora.get("energy", 0), tag + ": energy_consumption"


def store_episode(store, prefix, inst_idx, kind, rng_seed, actions, ref, full_states, mo=None):
    T = ref["T"]
    store[prefix + "inst"] = np.int32(inst_idx)
    store[prefix + "rng_seed"] = np.uint64(rng_seed)
    store[prefix + "actions"] = actions[:T].copy()
    for key, dt in (("k", np.int16), ("m", np.int16), ("job_r", np.int16), ("job_n", np.int32), ("done", np.uint8),
                    ("step_time", np.int32), ("delay", np.int64)):
        store[prefix + key] = ref[key].astype(dt)
    store[prefix + "reward"] = ref["reward"].astype(np.float64)
    store[prefix + "state0"] = ref["state0"]
    store[prefix + "tend"] = ref["tend"]
    store[prefix + "final"] = np.array([ref["makespan"], ref["delay_time_sum"], T], np.int64)
    store[prefix + "states_sha256"] = np.frombuffer(hashlib.sha256(bits(ref["states"]).tobytes()).digest(), np.uint8)
    store[prefix + "state_last"] = ref["states"][-1]
    store[prefix + "completion"] = np.int64(ref.get("completion_time", 0))
    store[prefix + "energy"] = np.int64(ref.get("energy", 0))
    if mo is not None and mo not in ("sf", "sod") and mo[0] == "dyn":   # [reward_policy, completion, tardiness, energy] (-1 = None)
        store[prefix + "mo"] = np.array([mo[1]] + [-1.0 if v is None else v for v in mo[2:5]], np.float64)
    elif mo is not None and mo not in ("sf", "sod"):
        store[prefix + "mo"] = np.array([mo[0], mo[1], -1.0 if mo[2] is None else mo[2], -1.0 if mo[3] is None else mo[3]], np.float64)
    if full_states:
        store[prefix + "states"] = ref["states"]


def store_instance(store, prefix, arr, name):
    store[prefix + "name"] = np.array(name)
    for key in ("Jr", "p", "elig_n", "elig_list", "count", "arrive", "delivery", "x"):
        store[prefix + key] = getattr(arr, key)
    store[prefix + "ddt"] = np.float64(arr.ddt)
    if hasattr(arr, "power"):
        for key in ("power", "idle_power", "bk_n", "bk"):
            store[prefix + key] = getattr(arr, key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None, help="generate only this suite (the others keep their files)")
    args = ap.parse_args()
    from environments.SO_FJSSP import SO_FJSSP_Environment
    from environments.MO_FJSSP_discretes import MO_FJSSP_Environment
    from environments.SO_SFJSP import SO_SFJSP_Environment
    from environments.MO_DFJSP_breakdown import MO_DFJSP_Environment
    from environments.SO_DFJSP import SO_DFJSP_Environment

    tmp = tempfile.mkdtemp(prefix="fjsp_golden_")
    report = []
    ALL_PAIRS = [("fixed", (a0, a1)) for a0 in range(6) for a1 in range(5)]

    def suite(name, cases, plans_store, plans_verify, full_state_eps, variant="so"):
        """cases: list of (label, arrays, folder_parent, folder_name)."""
        if args.only and args.only != name:
            return
        EnvCls = {"so": SO_FJSSP_Environment, "mo": MO_FJSSP_Environment, "sf": SO_SFJSP_Environment,
                  "dyn": MO_DFJSP_Environment, "sod": SO_DFJSP_Environment}[variant]
        store = {}
        suite_base = splitmix64(sum(ord(ch) for ch in name) * 7919)
        store["rng_seed_base"] = np.uint64(suite_base)
        n_eps = n_steps = 0
        timing = [0.0, 0]
        ep_id = 0
        for ci, (label, arr, parent, folder) in enumerate(cases):
            store_instance(store, "i%d_" % ci, arr, label)
            Tmax = int(sum(int(arr.count[s][r]) * int(arr.Jr[r]) for s in range(arr.S) for r in range(arr.R))) + 8
            checked_loader = False
            mo_memo = {}
            for plan in plans_store(ci) + [tuple(pl) + (False,) for pl in plans_verify(ci)]:
                if variant in ("so", "sf", "sod"):
                    (kind, seed, keep), mo = plan, {"so": None, "sf": "sf", "sod": "sod"}[variant]
                elif variant == "dyn":
                    # (kind, seed, reward_policy, keep); policy 3 takes its normalisers from the policy 0/1/2 runs
                    kind, seed, policy, keep = plan
                    mo = ("dyn", policy) + tuple(mo_memo.get(kk) if policy == 3 else None
                                                 for kk in ("completion", "tardiness", "energy"))
                else:
                    # (kind, seed, mo_spec, keep); mo_spec = (w0, w1, use_normalisers)
                    kind, seed, mo_spec, keep = plan
                    cn = mo_memo.get("completion") if mo_spec[2] else None
                    tn = mo_memo.get("tardiness") if mo_spec[2] else None
                    mo = (mo_spec[0], mo_spec[1], cn, tn)
                actions = action_stream(kind, seed, Tmax, {"so": None, "mo": 18, "sf": 20, "dyn": "dyn", "sod": None}[variant])
                # stored episode e of a suite plays with random.choice stream seed
                # suite_base + e * 1000003 == the seed env e of a batch created with
                # rng_seed = suite_base gets (fjsp_kernels.hip bind()); verify-only
                # episodes use an unrelated seed
                rng_seed = (suite_base + ep_id * 1000003) & MASK64 if keep else splitmix64(seed * 1000003 + ci)
                ref, env = run_reference(EnvCls, arr, parent, folder, actions, rng_seed,
                                         check_lp=(n_eps % 16 == 0), timing=timing, mo=mo)
                if variant == "dyn" and mo[1] != 3:   # noqa: E129
                    mo_memo[("completion", "tardiness", "energy")[mo[1]]] = (ref["completion_time"], ref["delay_time_sum"],
                                                                              ref["energy"])[mo[1]]
                if variant == "mo":      # MPPPO.py:161-164: the single-objective runs supply the normalisers
                    if mo[0] == 1 and mo[2] is None:
                        mo_memo["completion"] = ref["completion_time"]
                    if mo[1] == 1 and mo[2] is None:
                        mo_memo["tardiness"] = ref["delay_time_sum"]
                if not checked_loader:
                    check_loader_against_reference(arr, env)
                    checked_loader = True
                ora = run_oracle(arr, actions, rng_seed, ref["T"], mo=mo)
                compare(ref, ora, "%s/%s/%s/%s" % (name, label, kind, seed))
                n_eps += 1; n_steps += ref["T"]
                if keep:
                    store_episode(store, "e%d_" % ep_id, ci, kind, rng_seed, actions, ref, ep_id in full_state_eps, mo)
                    ep_id += 1
        store["n_instances"] = np.int32(len(cases))
        store["n_episodes"] = np.int32(ep_id)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **store)
        line = ("%-12s instances %3d  episodes verified %4d (stored %3d)  steps %6d  reference %.0f steps/s (1 core)  %.0f KB"
                % (name, len(cases), n_eps, ep_id, n_steps, timing[1] / max(timing[0], 1e-9), os.path.getsize(path) / 1024))
        print(line); report.append(line)

    # ---- suite 1: Brandimarte Mk01 (BASELINE config 1) -------------------------
    s1 = fi.InstanceSet(1).load_csv(0, REF + "/data/benchmark/Brandimarte_Data", "Mk01").solve_fluid()
    suite("mk01", [("Mk01", s1.arrays(0), REF + "/data/benchmark/Brandimarte_Data", "Mk01")],
          lambda ci: [(kp, 0, True) for kp in ALL_PAIRS] + [(("random",), sd, True) for sd in (11, 12, 13, 14)],
          lambda ci: [] if args.quick else [(("random",), sd) for sd in range(100, 116)],
          full_state_eps={0, 7, 30})

    # ---- suite 2: synthetic 10x5 (BASELINE config 2 generator, seeds 1000+i) ----
    n_syn = 8 if args.quick else 48
    s2 = fi.InstanceSet(n_syn).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    cases = []
    for i in range(n_syn):
        arr = s2.arrays(i)
        write_csv_folder(arr, os.path.join(tmp, "syn", "S%04d" % i))
        cases.append(("seed%d" % (1000 + i), arr, os.path.join(tmp, "syn"), "S%04d" % i))
    suite("synth10x5", cases,
          lambda ci: ([(kp, 0, True) for kp in ALL_PAIRS[(ci * 7) % 30::6]] + [(("random",), 500 + ci, True)]) if ci < 16 else [],
          lambda ci: [(kp, 0) for kp in ALL_PAIRS] + [(("random",), 900 + ci), (("random",), 1900 + ci)],
          full_state_eps={0, 1})

    # ---- suite 3: multi-job list semantics --------------------
    s3 = fi.InstanceSet(4)
    s3.load_csv(0, REF + "/data/DDQN", "P11").load_csv(1, REF + "/data/MPPPO", "DDT0.5_M10_R5")
    s3.load_csv(2, REF + "/data/DDQN", "P21")
    s3.generate(3, 4242, fi.GenParams(R_min=4, R_max=4, J_min=3, J_max=4, M=6, p_min=5, p_max=60, N_min=2, N_max=6,
                                      S=1, DDT=1.0, t_si_min=100.0, t_si_max=200.0))
    s3.solve_fluid()
    write_csv_folder(s3.arrays(3), os.path.join(tmp, "mj", "G4242"))
    cases = [("DDQN/P11", s3.arrays(0), REF + "/data/DDQN", "P11"),
             ("MPPPO/DDT0.5_M10_R5", s3.arrays(1), REF + "/data/MPPPO", "DDT0.5_M10_R5"),
             ("DDQN/P21", s3.arrays(2), REF + "/data/DDQN", "P21"),
             ("gen4242", s3.arrays(3), os.path.join(tmp, "mj"), "G4242")]
    suite("multijob", cases,
          lambda ci: [(("random",), 21 + ci, True), (("fixed", (0, 0)), 0, True), (("fixed", (2, 3)), 0, True),
                      (("fixed", (4, 1)), 0, True)],
          lambda ci: [] if args.quick else [(kp, 0) for kp in ALL_PAIRS[::3]] + [(("random",), 77 + ci)],
          full_state_eps=set())

    # ---- suite 4: large instances (K > 64 -> multi-chunk kernels; M > 8 -> CPython set order) ------
    big = [("benchmark/Brandimarte_Data", "Mk04"), ("benchmark/Brandimarte_Data", "Mk06"),
           ("benchmark/Brandimarte_Data", "Mk10"), ("MPPPO", "DDT1.0_M15_R10"), ("MPPPO", "DDT0.5_M20_R5"),
           ("DDQN", "P83")]
    if args.quick:
        big = big[:2]
    s4 = fi.InstanceSet(len(big))
    for i, (d, f) in enumerate(big):
        s4.load_csv(i, REF + "/data/" + d, f)
    s4.solve_fluid()
    cases = [(d.split("/")[-1] + "/" + f, s4.arrays(i), REF + "/data/" + d, f) for i, (d, f) in enumerate(big)]
    suite("large", cases,
          lambda ci: [(("random",), 31 + ci, True), (("fixed", (3, 3)), 0, True), (("fixed", (0, 4)), 0, True)],
          lambda ci: [] if args.quick else [(("fixed", (2, 0)), 0), (("fixed", (5, 1)), 0), (("random",), 131 + ci)],
          full_state_eps=set())

    # ---- suite 4a: degenerate sizes (one kind / one operation / one machine / several jobs of one kind) ----
    edge_prm = [dict(R_min=1, R_max=1, J_min=1, J_max=1, M=1, N_min=1, N_max=1),
                dict(R_min=1, R_max=1, J_min=1, J_max=1, M=2, N_min=3, N_max=3),
                dict(R_min=1, R_max=1, J_min=3, J_max=3, M=1, N_min=2, N_max=2),
                dict(R_min=2, R_max=2, J_min=1, J_max=2, M=2, N_min=1, N_max=2),
                dict(R_min=3, R_max=3, J_min=2, J_max=2, M=8, N_min=1, N_max=1),
                dict(R_min=2, R_max=2, J_min=5, J_max=5, M=3, N_min=4, N_max=5)]
    s4a = fi.InstanceSet(len(edge_prm))
    cases = []
    for i, kw in enumerate(edge_prm):
        s4a.generate(i, 9000 + i, fi.GenParams(p_min=1, p_max=9, S=1, DDT=1.0, t_si_min=100.0, t_si_max=200.0, **kw))
    s4a.solve_fluid()
    for i in range(len(edge_prm)):
        write_csv_folder(s4a.arrays(i), os.path.join(tmp, "edge", "E%d" % i))
        cases.append(("edge%d" % i, s4a.arrays(i), os.path.join(tmp, "edge"), "E%d" % i))
    suite("edge", cases,
          lambda ci: [(kp, 0, True) for kp in ALL_PAIRS[ci % 3::3]] + [(("random",), 7 + ci, True)],
          lambda ci: [(kp, 0) for kp in ALL_PAIRS],
          full_state_eps=set(range(0, 66, 5)))

    # ---- suite 4b: multi-order instances (order arrival re-solves the LP mid-episode, SO_FJSSP.py:218-231).
    # Oracle-only for now: the kernels reject S > 1.
    s4b = fi.InstanceSet(3)
    s4b.load_csv(0, REF + "/data/HMPSAC", "DDT0.5_M20_S3")
    s4b.generate(1, 777, fi.GenParams(R_min=3, R_max=3, J_min=2, J_max=3, M=4, p_min=20, p_max=90, N_min=2, N_max=4,
                                      S=3, DDT=1.0, t_si_min=100.0, t_si_max=200.0))
    s4b.generate(2, 778, fi.GenParams(R_min=4, R_max=4, J_min=3, J_max=4, M=5, p_min=40, p_max=400, N_min=2, N_max=3,
                                      S=4, DDT=0.5, t_si_min=100.0, t_si_max=200.0))
    s4b.solve_fluid()
    for i in (1, 2):
        write_csv_folder(s4b.arrays(i), os.path.join(tmp, "mo_ord", "G%d" % i))
    cases = [("HMPSAC/DDT0.5_M20_S3", s4b.arrays(0), REF + "/data/HMPSAC", "DDT0.5_M20_S3"),
             ("gen777", s4b.arrays(1), os.path.join(tmp, "mo_ord"), "G1"),
             ("gen778", s4b.arrays(2), os.path.join(tmp, "mo_ord"), "G2")]
    suite("multiorder", cases,
          lambda ci: [(("random",), 301 + ci, True), (("fixed", (2, 0)), 0, True), (("fixed", (4, 2)), 0, True)],
          lambda ci: [] if args.quick else [(("fixed", (0, 3)), 0), (("fixed", (3, 1)), 0), (("random",), 401 + ci)],
          full_state_eps=set())

    # ---- suite 5: MO_FJSSP_discretes (the environment agents/MPPPO/MPPPO.py instantiates) ---------
    s5g = fi.InstanceSet(3).generate_range(1000, fi.bench_10x5_params())
    mo_cases = [("benchmark/Brandimarte_Data", "Mk01"), ("MPPPO", "DDT0.5_M10_R5"), ("MPPPO", "DDT1.5_M15_R5")]
    s5 = fi.InstanceSet(len(mo_cases) + 3)
    cases = []
    for i, (d, f) in enumerate(mo_cases):
        s5.load_csv(i, REF + "/data/" + d, f)
    for i in range(3):                                  # reload through the CSV reader so DDT parses like the reference's
        write_csv_folder(s5g.arrays(i), os.path.join(tmp, "mo", "S%d" % i))
        s5.load_csv(len(mo_cases) + i, os.path.join(tmp, "mo"), "S%d" % i)
    s5.solve_fluid()
    for i, (d, f) in enumerate(mo_cases):
        cases.append((d.split("/")[-1] + "/" + f, s5.arrays(i), REF + "/data/" + d, f))
    for i in range(3):
        cases.append(("seed%d" % (1000 + i), s5.arrays(len(mo_cases) + i), os.path.join(tmp, "mo"), "S%d" % i))
    suite("mo_discretes", cases,
          lambda ci: [(("random",), 41 + ci, (1, 0, False), True), (("random",), 51 + ci, (0, 1, False), True),
                      (("random",), 61 + ci, (0.5, 0.5, True), True), (("random",), 71 + ci, (0.75, 0.25, True), True),
                      (("fixed", (4, 0)), 0, (0, 1, False), True), (("fixed", (16, 0)), 0, (0.25, 0.75, True), True)],
          lambda ci: [] if args.quick else [(("fixed", (a, 0)), 0, (0.5, 0.5, True)) for a in range(18)],
          full_state_eps={0, 2}, variant="mo")

    # ---- suite 6: SO_SFJSP (the environment agents/DDQN/DDQN.py instantiates): makespan reward ------
    sf_cases = [("benchmark/Brandimarte_Data", "Mk01"), ("DDQN", "P11"), ("DDQN", "P41"), ("MPPPO", "DDT0.5_M10_R5")]
    s6 = fi.InstanceSet(len(sf_cases) + 2)
    for i, (d, f) in enumerate(sf_cases):
        s6.load_csv(i, REF + "/data/" + d, f)
    s6g = fi.InstanceSet(2).generate_range(1000, fi.bench_10x5_params())
    for i in range(2):
        write_csv_folder(s6g.arrays(i), os.path.join(tmp, "sf", "S%d" % i))
        s6.load_csv(len(sf_cases) + i, os.path.join(tmp, "sf"), "S%d" % i)
    s6.solve_fluid()
    cases = [(d.split("/")[-1] + "/" + f, s6.arrays(i), REF + "/data/" + d, f) for i, (d, f) in enumerate(sf_cases)]
    cases += [("seed%d" % (1000 + i), s6.arrays(len(sf_cases) + i), os.path.join(tmp, "sf"), "S%d" % i) for i in range(2)]
    suite("so_sfjsp", cases,
          lambda ci: [(("random",), 81 + ci, True), (("random",), 91 + ci, True), (("fixed", (6, 0)), 0, True),
                      (("fixed", (13, 0)), 0, True)],
          lambda ci: [] if args.quick else [(("fixed", (a, 0)), 0) for a in range(20)],
          full_state_eps={0}, variant="sf")

    # ---- suite 7: MO_DFJSP_breakdown (BASELINE config 5): order arrivals + machine breakdowns + energy ------
    dyn_cases = [("industrial", "DDT0.5_M20_S1"), ("industrial", "DDT0.5_M20_S3"), ("industrial", "DDT0.5_M20_S5"),
                 ("HMPSAC", "DDT0.5_M10_S1"), ("HMPSAC", "DDT1.0_M15_S3")]
    if args.quick:
        dyn_cases = dyn_cases[:2]
    n_gen = 4
    s7 = fi.InstanceSet(len(dyn_cases) + n_gen)
    for i, (d, f) in enumerate(dyn_cases):
        s7.load_csv(i, REF + "/data/" + d, f)
    gen_prm = [dict(R_min=3, R_max=3, J_min=2, J_max=3, M=4, p_min=5, p_max=30, N_min=2, N_max=3, S=2, DDT=1.0),
               dict(R_min=4, R_max=4, J_min=3, J_max=4, M=6, p_min=10, p_max=60, N_min=1, N_max=3, S=3, DDT=0.5),
               dict(R_min=2, R_max=2, J_min=2, J_max=2, M=3, p_min=1, p_max=6, N_min=2, N_max=2, S=1, DDT=1.0),
               dict(R_min=5, R_max=5, J_min=3, J_max=5, M=10, p_min=20, p_max=200, N_min=2, N_max=4, S=2, DDT=1.5)]
    s7g = fi.InstanceSet(n_gen)
    rs = np.random.RandomState(2024)
    for i, kw in enumerate(gen_prm):
        s7g.generate(i, 8800 + i, fi.GenParams(t_si_min=100.0, t_si_max=200.0, **kw))
        a = s7g.arrays(i)
        power = np.where(a.p > 0, rs.randint(1, 50, a.p.shape), 0)
        idle = rs.randint(1, 10, a.M)
        horizon = int(a.p.max()) * int(a.count.sum()) * int(a.Jr.max()) // a.M + 50
        bk_n = rs.randint(0, 5, a.M)
        bk = []
        for m in range(a.M):                            # sorted disjoint windows; dense enough that every branch
            t = 0                                       # of MO_DFJSP_breakdown.py:204-231 is taken
            for _ in range(int(bk_n[m])):
                st = t + int(rs.randint(1, max(2, horizon // 6)))
                en = st + int(rs.randint(1, max(2, int(a.p.max()))))
                bk.append((st, en)); t = en
        s7g.set_dynamic(i, power, idle, bk_n, np.array(bk, np.int32).reshape(-1, 2))
        write_csv_folder(s7g.arrays(i), os.path.join(tmp, "dyn", "D%d" % i))
        s7.load_csv(len(dyn_cases) + i, os.path.join(tmp, "dyn"), "D%d" % i)
    s7.solve_fluid()
    cases = [(d + "/" + f, s7.arrays(i), REF + "/data/" + d, f) for i, (d, f) in enumerate(dyn_cases)]
    cases += [("gen%d" % (8800 + i), s7.arrays(len(dyn_cases) + i), os.path.join(tmp, "dyn"), "D%d" % i) for i in range(n_gen)]
    DYN_PAIRS = [("fixed", (a0, a1)) for a0 in range(12) for a1 in range(10)]
    suite("mo_dfjsp", cases,
          lambda ci: [(("random",), 601 + ci, 0, True), (("random",), 611 + ci, 1, True), (("random",), 621 + ci, 2, True),
                      (("random",), 631 + ci, 3, True), (DYN_PAIRS[(ci * 37) % 120], 0, 3, True),
                      (DYN_PAIRS[(ci * 53 + 67) % 120], 0, 2, True)],
          lambda ci: ([] if args.quick else [(kp, 0, 3) for kp in (DYN_PAIRS if ci >= len(dyn_cases) else DYN_PAIRS[ci::7])]),
          full_state_eps={0, 3}, variant="dyn")

    # ---- suite 8: SO_DFJSP (agents/DA3C's environment): SO_FJSSP over class_FJSP.py (due date = order delivery) ------
    sod_cases = [("benchmark/Brandimarte_Data", "Mk01"), ("DDQN", "P11"), ("HMPSAC", "DDT0.5_M10_S1"), ("HMPSAC", "DDT1.0_M15_S3")]
    if args.quick:
        sod_cases = sod_cases[:2]
    s8 = fi.InstanceSet(len(sod_cases) + 3)
    for i, (d, f) in enumerate(sod_cases):
        s8.load_csv(i, REF + "/data/" + d, f)
    sod_prm = [dict(R_min=3, R_max=3, J_min=2, J_max=3, M=4, p_min=20, p_max=90, N_min=2, N_max=4, S=3, DDT=1.0),
               dict(R_min=4, R_max=4, J_min=3, J_max=4, M=9, p_min=5, p_max=60, N_min=1, N_max=3, S=2, DDT=0.5),
               dict(R_min=2, R_max=2, J_min=1, J_max=2, M=3, p_min=1, p_max=9, N_min=1, N_max=2, S=1, DDT=1.0)]
    s8g = fi.InstanceSet(3)
    for i, kw in enumerate(sod_prm):
        seed = 6600 + i
        s8g.generate(i, seed, fi.GenParams(t_si_min=100.0, t_si_max=200.0, **kw))
        while not (s8g.arrays(i).p > 0).any(axis=0).all():        # class_FJSP.py:159 divides by len(kind_task_tuple)
            seed += 7919
            s8g.generate(i, seed, fi.GenParams(t_si_min=100.0, t_si_max=200.0, **kw))
        write_csv_folder(s8g.arrays(i), os.path.join(tmp, "sod", "G%d" % i))
        s8.load_csv(len(sod_cases) + i, os.path.join(tmp, "sod"), "G%d" % i)
    s8.solve_fluid()
    cases = [(d.split("/")[-1] + "/" + f, s8.arrays(i), REF + "/data/" + d, f) for i, (d, f) in enumerate(sod_cases)]
    cases += [("gen%d" % (6600 + i), s8.arrays(len(sod_cases) + i), os.path.join(tmp, "sod"), "G%d" % i) for i in range(3)]
    suite("so_dfjsp", cases,
          lambda ci: [(("random",), 701 + ci, True), (("fixed", (3, 3)), 0, True), (("fixed", (0, 0)), 0, True),
                      (("fixed", (4, 2)), 0, True)],
          lambda ci: [] if args.quick else [(kp, 0) for kp in ALL_PAIRS[ci % 2::2]] + [(("random",), 801 + ci)],
          full_state_eps={0}, variant="sod")

    report.append("LP checks vs HiGHS on the reference-built model: %d solves, max objective gap %.2e, max infeasibility %.2e"
                  % (LP_STATS["solves"], LP_STATS["max_obj_gap"], LP_STATS["max_infeas"]))
    print(report[-1])
    with open(os.path.join(HERE, "GENERATION_REPORT.txt"), "a" if args.only else "w") as f:
        if not args.only:
            f.write("make_golden.py: every episode below was played on the reference (under oracle/ref_shim)\n"
                    "and on the C oracle and compared bit-exactly (trace, rewards, f64 states).\n\n")
        f.write("\n".join(report) + "\n")


if __name__ == "__main__":
    main()
