"""Shared test helpers: golden-fixture access and oracle drivers (test infrastructure)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUITES = ("mk01", "synth10x5", "multijob", "large", "edge", "multiorder")      # SO_FJSSP (multiorder: S > 1)
SOD_SUITES = ("so_dfjsp",)                                   # SO_DFJSP (SO_FJSSP over class_FJSP.py: due date = order delivery)
MO_SUITES = ("mo_discretes",)                                # MO_FJSSP_discretes
SF_SUITES = ("so_sfjsp",)                                    # SO_SFJSP
DYN_SUITES = ("mo_dfjsp",)                                   # MO_DFJSP_breakdown (order arrivals, breakdowns, energy)
ORACLE_ONLY_SUITES = ()

# observation entries that pass through math.pow(x, 2) + sqrt in the reference
# (SO_FJSSP.py:86-95): glibc pow differs from x*x by 1 ulp in ~0.08 % of arguments,
# so the kernels (x*x) are compared with a tolerance there and bit-exactly elsewhere.
POW_OBS = (1, 3, 5)
POW_COLS = POW_OBS + tuple(10 + i for i in POW_OBS)
EXACT_COLS = tuple(i for i in range(20) if i not in POW_COLS)
# MO_FJSSP_discretes state = [7 static | 9 obs | 9 deltas]; pow()-derived: static N_std, J_std
# (MO_FJSSP_discretes.py:59-63) and obs ct_std, cro_std, gap_std (:70-80)
MO_POW_COLS = (4, 6, 7, 9, 11, 16, 18, 20)
MO_EXACT_COLS = tuple(i for i in range(25) if i not in MO_POW_COLS)
# SO_SFJSP state = [9 obs | 9 deltas]; pow()-derived: ct_std, cro_std, gap_std, gap_m_std (SO_SFJSP.py:68-81)
SF_POW_COLS = (1, 3, 6, 8, 10, 12, 15, 17)
SF_EXACT_COLS = tuple(i for i in range(18) if i not in SF_POW_COLS)
# MO_DFJSP_breakdown state = [15 obs | 15 deltas]; pow()-derived: ct_std, cro_std, gap_std, gap_m_std (:103-116)
DYN_POW_COLS = (3, 6, 8, 10, 18, 21, 23, 25)
DYN_EXACT_COLS = tuple(i for i in range(30) if i not in DYN_POW_COLS)
# The standard-deviation entries: the reference squares with math.pow (glibc's pow differs from x * x by 1 ulp in ~0.08 % of
# arguments) and adds the squares left to right; the kernels square with x * x and add them by a fixed tree (csrc/fjsp_common.h
# row_tree_sum_f64: no sequential walk for entries that feed no decision).  Both effects are below 1e-14 relative.
POW_RTOL = 1e-11   # north_star allows 1e-5
POW_ATOL = 1e-12   # the v(t) - v(t-1) half cancels, so an absolute floor is needed


class Arrays(object):
    pass


def load_suite(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    insts = []
    for i in range(int(z["n_instances"])):
        a = Arrays()
        for key in ("Jr", "p", "elig_n", "elig_list", "count", "arrive", "delivery", "x"):
            setattr(a, key, z["i%d_%s" % (i, key)])
        for key in ("power", "idle_power", "bk_n", "bk"):
            if "i%d_%s" % (i, key) in z.files:
                setattr(a, key, z["i%d_%s" % (i, key)])
        a.ddt = float(z["i%d_ddt" % i])
        a.name = str(z["i%d_name" % i])
        a.R, a.S = len(a.Jr), len(a.arrive)
        a.K, a.M = a.p.shape
        insts.append(a)
    eps = []
    for e in range(int(z["n_episodes"])):
        d = {key: z["e%d_%s" % (e, key)] for key in
             ("inst", "rng_seed", "actions", "k", "m", "job_r", "job_n", "done", "step_time", "delay", "reward",
              "state0", "tend", "final", "states_sha256", "state_last")}
        if "e%d_states" % e in z.files:
            d["states"] = z["e%d_states" % e]
        for opt in ("mo", "completion", "energy"):
            if "e%d_%s" % (e, opt) in z.files:
                d[opt] = z["e%d_%s" % (e, opt)]
        d["inst"] = int(d["inst"]); d["rng_seed"] = int(d["rng_seed"]); d["T"] = int(d["final"][2])
        eps.append(d)
    return insts, eps, int(z["rng_seed_base"])


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def states_digest(states):
    return np.frombuffer(hashlib.sha256(bits(states).tobytes()).digest(), np.uint8)


def instance_set_from(arrs):
    """Rebuild a product InstanceSet (with the stored fluid solution) from fixture arrays."""
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    s = fi.InstanceSet(len(arrs))
    for i, a in enumerate(arrs):
        s.set_raw(i, a.Jr, a.p, a.elig_n, a.elig_list, a.count, a.arrive, a.delivery, a.ddt)
        s.set_x(i, a.x)
        if hasattr(a, "power"):
            s.set_dynamic(i, a.power, a.idle_power, a.bk_n, a.bk)
    return s


def play_oracle(arr, x, actions, rng_seed, variant=0, mo=None):
    """Play one episode on the C oracle; returns a dict shaped like the fixtures.
    mo = (w0, w1, completion, tardiness) with <= 0 standing for None selects the MO variant's step."""
    from oracle import pyoracle
    if arr.S > 1:        # order arrivals re-solve the LP on the live state (class_FJSSP.py:239): product LP as the hook
        from deep_reinforcement_learning_for_fjsp_amd import instances as fi
        x = lambda Q, now: fi.fluid_lp(arr.Jr, arr.p, Q, now)[0]
    env = pyoracle.OracleEnv(arr, x, variant, rng_seed, ddt=getattr(arr, "ddt", None) if variant in (2, 4) else None)
    rec = {k: [] for k in ("k", "m", "job_r", "job_n", "reward", "done", "step_time", "delay", "states")}
    state0 = env.reset()
    t = 0
    while not env.done:
        if mo is None and variant == 1:
            s, r, d = env.step_sf(int(actions[t][0]))
        elif variant == 4:   # mo = (reward_policy, completion, tardiness, energy), <= 0 standing for None
            s, r, d = env.step_dyn(actions[t], int(mo[0]), *[v if v > 0 else None for v in mo[1:4]])
        elif mo is None:
            s, r, d = env.step(actions[t])
        else:
            s, r, d = env.step_mo(int(actions[t][0]), (mo[0], mo[1]), mo[2] if mo[2] > 0 else None,
                                  mo[3] if mo[3] > 0 else None)
        tr = env.trace
        rec["k"].append(tr.k_sel); rec["m"].append(tr.m_sel); rec["job_r"].append(tr.job_kind)
        rec["job_n"].append(tr.job_n); rec["reward"].append(r); rec["done"].append(d)
        rec["step_time"].append(tr.step_time); rec["delay"].append(tr.delay_time_sum); rec["states"].append(s)
        t += 1
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(state0=state0, tend=env.machine_time_end(), makespan=env.makespan, delay_time_sum=env.delay_time_sum,
               T=t, fluid_completed_time=env.fluid_completed_time, completion_time=env.completion_time)
    if variant == 4:
        out["energy"] = env.energy_consumption
    return out


def assert_state_close(got, want, what="", mo=False, sf=False, dyn=False):
    """Kernel state vs oracle/reference state: bit-exact except the pow()-derived entries."""
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    ex = list(DYN_EXACT_COLS if dyn else (SF_EXACT_COLS if sf else (MO_EXACT_COLS if mo else EXACT_COLS)))
    if not np.array_equal(bits(got[..., ex]), bits(want[..., ex])):
        bad = np.argwhere(bits(got[..., ex]) != bits(want[..., ex]))[0]
        raise AssertionError("%s exact state entry differs at %s: got %r want %r"
                             % (what, bad, got[..., ex][tuple(bad)], want[..., ex][tuple(bad)]))
    pw = list(DYN_POW_COLS if dyn else (SF_POW_COLS if sf else (MO_POW_COLS if mo else POW_COLS)))
    # A pow()-derived entry of the second half of the state is the DIFFERENCE of two such observations (v(t) - v(t-1)):
    # a last-bit difference in either operand survives the cancellation at the operands' magnitude (clocks beyond 2^22:
    # std of machine end times ~2e4, difference ~1 -> 4e-12 relative), so those entries get an absolute tolerance of a
    # few ulp of the observation they are the difference of.
    shift = 15 if dyn else (9 if (sf or mo) else 10)
    first_delta = (15 if dyn else (9 if sf else (16 if mo else 10)))
    for c in pw:
        atol = POW_ATOL
        if c >= first_delta and (c - shift) in pw:
            atol = max(atol, 2e-14 * float(np.max(np.abs(want[..., c - shift]))) if want.size else atol)
        np.testing.assert_allclose(got[..., c], want[..., c], rtol=POW_RTOL, atol=atol, err_msg="%s (state entry %d)" % (what, c))


import contextlib


@contextlib.contextmanager
def env_var(name, value):
    """Set (or with None: leave) an environment variable for the duration of a block."""
    old = os.environ.get(name)
    if value is not None:
        os.environ[name] = value
    try:
        yield
    finally:
        if value is not None:
            if old is None:
                del os.environ[name]
            else:
                os.environ[name] = old
