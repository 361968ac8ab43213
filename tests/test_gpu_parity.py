"""GPU parity: the HIP path (through the C ABI) against the golden reference
traces and against the C oracle, on the same instances, fluid solution, actions
and random.choice stream.

Bars: operation/machine choices, rewards, done flags, clocks, tardiness,
makespan and machine completion times BIT-EXACT; state vectors bit-exact except
the three math.pow()-derived entries (tests/helpers.py POW_COLS), which are held
to 1e-12 (the north star allows 1e-5).
"""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _batch_for(suite, torch):
    """One env per stored episode: env e <- (instance of episode e, rng stream of episode e)."""
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
    insts, eps, base = H.load_suite(suite)
    s = H.instance_set_from([insts[ep["inst"]] for ep in eps])
    b = EnvBatch(s, len(eps), rng_seed=base, variant=5 if suite in H.SOD_SUITES else 0)
    for e, ep in enumerate(eps):
        assert b.env_seed(e) == ep["rng_seed"]
    T = max(ep["T"] for ep in eps)
    actions = np.zeros((T, len(eps), 2), np.uint8)
    for e, ep in enumerate(eps):
        actions[:ep["T"], e] = ep["actions"]
    return insts, eps, b, torch.from_numpy(actions).to(b.device), T


@pytest.mark.parametrize("suite", H.SUITES)
def test_fluid_tables_match_oracle(torch_gpu, suite):
    """update_fluid_parameter on the device == the oracle's tables (class_FJSSP.py:282-306), bit-exact."""
    from oracle import pyoracle
    insts, eps, b, _, _ = _batch_for(suite, torch_gpu)
    seen = set()
    for e, ep in enumerate(eps):
        if ep["inst"] in seen:
            continue
        seen.add(ep["inst"])
        a = insts[ep["inst"]]
        env = pyoracle.OracleEnv(a, a.x)
        env.reset()
        want = env.fluid_tables()
        got = b.fluid_tables(e)
        for g, w, name in zip(got, want, ("rate", "arr", "rate_sum", "time_sum")):
            assert np.array_equal(H.bits(g), H.bits(w)), (suite, a.name, name)


@pytest.mark.parametrize("suite", H.SUITES + H.SOD_SUITES)
def test_step_kernel_matches_reference_fixtures(torch_gpu, suite):
    """Per-step launches through fjsp_env_step vs the stored reference traces."""
    torch = torch_gpu
    insts, eps, b, actions, T = _batch_for(suite, torch)
    st0 = b.reset().cpu().numpy()
    for e, ep in enumerate(eps):
        H.assert_state_close(st0[e], ep["state0"], "%s ep %d reset" % (suite, e))
    states = np.zeros((T, len(eps), 20)); rewards = np.zeros((T, len(eps))); dones = np.zeros((T, len(eps)), np.uint8)
    clocks = np.zeros((T, len(eps)), np.int64); delays = np.zeros((T, len(eps)), np.int64)
    km = np.zeros((T, len(eps), 2), np.int16)
    tr = torch.zeros(len(eps), 2, dtype=torch.int16, device=b.device)
    for t in range(T):
        s, r, d = b.step(actions[t], trace_out=tr)
        rd = b.read()
        states[t] = s.cpu().numpy(); rewards[t] = r.cpu().numpy(); dones[t] = d.cpu().numpy()
        km[t] = tr.cpu().numpy()
        clocks[t] = rd["step_time"].cpu().numpy(); delays[t] = rd["delay_time_sum"].cpu().numpy()
    final = b.read()
    tend = b.machine_time_end().cpu().numpy()
    status = final["status"].cpu().numpy()
    for e, ep in enumerate(eps):
        Te = ep["T"]
        tag = "%s episode %d (%s)" % (suite, e, insts[ep["inst"]].name)
        # what the rule pair resolved to, step by step (task_select / machine_select, SO_FJSSP.py:173-174)
        assert np.array_equal(km[:Te, e, 0], ep["k"]), tag + " chosen operation type"
        assert np.array_equal(km[:Te, e, 1], ep["m"]), tag + " chosen machine"
        assert (km[Te:, e] == -1).all(), tag + " (k, m) of an env that did not step"
        assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(ep["reward"])), tag + " reward"
        assert np.array_equal(dones[:Te, e], ep["done"]), tag + " done"
        assert np.array_equal(clocks[:Te, e], ep["step_time"]), tag + " step_time"
        assert np.array_equal(delays[:Te, e], ep["delay"]), tag + " delay_time_sum"
        if "states" in ep:
            H.assert_state_close(states[:Te, e], ep["states"], tag)
        H.assert_state_close(states[Te - 1, e], ep["state_last"], tag + " last state")
        M = insts[ep["inst"]].M
        assert np.array_equal(tend[e, :M], ep["tend"]), tag + " machine time_end"
        assert int(final["makespan"][e]) == ep["final"][0] and int(final["delay_time_sum"][e]) == ep["final"][1], tag
        assert int(final["step_count"][e]) == Te and int(final["done"][e]) == 1
        # only the "stepped after done" bit may be set (episodes shorter than T keep receiving actions)
        assert status[e] & ~4 == 0, tag + " status %d" % status[e]
        assert (status[e] & 4 != 0) == (Te < T)


@pytest.mark.parametrize("suite", H.SUITES + H.SOD_SUITES)
def test_rollout_kernel_matches_reference_fixtures(torch_gpu, suite):
    """The fused T-step kernel: chosen (operation, machine) per step, rewards, final attributes."""
    torch = torch_gpu
    insts, eps, b, actions, T = _batch_for(suite, torch)
    b.reset()
    trace, rewards, state = b.rollout(actions)
    trace = trace.cpu().numpy(); rewards = rewards.cpu().numpy(); state = state.cpu().numpy()
    final = b.read()
    tend = b.machine_time_end().cpu().numpy()
    for e, ep in enumerate(eps):
        Te = ep["T"]
        tag = "%s episode %d (%s)" % (suite, e, insts[ep["inst"]].name)
        assert np.array_equal(trace[:Te, e, 0], ep["k"]), tag + " chosen operation type"
        assert np.array_equal(trace[:Te, e, 1], ep["m"]), tag + " chosen machine"
        assert (trace[Te:, e] == -1).all(), tag
        assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(ep["reward"])), tag + " reward"
        assert (rewards[Te:, e] == 0).all()
        H.assert_state_close(state[e], ep["state_last"], tag + " last state")
        M = insts[ep["inst"]].M
        assert np.array_equal(tend[e, :M], ep["tend"]), tag
        assert int(final["makespan"][e]) == ep["final"][0] and int(final["delay_time_sum"][e]) == ep["final"][1], tag
        assert int(final["status"][e]) == 0 and int(final["done"][e]) == 1 and int(final["step_count"][e]) == Te


def test_mo_discretes_matches_reference_fixtures(torch_gpu):
    """MO_FJSSP_discretes (what MPPPO instantiates): per-step and fused kernels vs the reference traces,
    with the reference's weight vectors / normalisers (MO_FJSSP_discretes.py:88,232-244)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_FJSSP_DISCRETES
    insts, eps, base = H.load_suite("mo_discretes")
    s = H.instance_set_from([insts[ep["inst"]] for ep in eps])
    T = max(ep["T"] for ep in eps)
    actions = np.zeros((T, len(eps), 2), np.uint8)
    for e, ep in enumerate(eps):
        actions[:ep["T"], e] = ep["actions"]
    actions = torch.from_numpy(actions).cuda()
    mo = torch.from_numpy(np.stack([ep["mo"] for ep in eps])).cuda()
    for mode in ("step", "rollout"):
        b = EnvBatch(s, len(eps), variant=VARIANT_MO_FJSSP_DISCRETES, rng_seed=base)
        assert b.state_size == 25
        st0 = b.reset().cpu().numpy()
        for e, ep in enumerate(eps):
            H.assert_state_close(st0[e], ep["state0"], "mo ep %d reset" % e, mo=True)
        if mode == "step":
            rewards = np.zeros((T, len(eps))); states = np.zeros((T, len(eps), 25))
            for t in range(T):
                st, r, d = b.step(actions[t], mo=mo)
                rewards[t] = r.cpu().numpy(); states[t] = st.cpu().numpy()
            trace = None
        else:
            trace, rw, st = b.rollout(actions, mo=mo)
            trace = trace.cpu().numpy(); rewards = rw.cpu().numpy(); last = st.cpu().numpy()
        fin = {k: v.cpu().numpy() for k, v in b.read().items()}
        tend = b.machine_time_end().cpu().numpy()
        for e, ep in enumerate(eps):
            Te = ep["T"]
            tag = "mo_discretes %s episode %d (%s)" % (mode, e, insts[ep["inst"]].name)
            # weighted rewards go through f64 divisions in the same order -> bit-exact too
            assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(ep["reward"])), tag + " reward"
            if trace is not None:
                assert np.array_equal(trace[:Te, e, 0], ep["k"]) and np.array_equal(trace[:Te, e, 1], ep["m"]), tag
                H.assert_state_close(last[e], ep["state_last"], tag, mo=True)
            else:
                H.assert_state_close(states[Te - 1, e], ep["state_last"], tag, mo=True)
                if "states" in ep:
                    H.assert_state_close(states[:Te, e], ep["states"], tag, mo=True)
            M = insts[ep["inst"]].M
            assert np.array_equal(tend[e, :M], ep["tend"]), tag
            assert fin["makespan"][e] == ep["final"][0] and fin["delay_time_sum"][e] == ep["final"][1], tag
            assert fin["completion_time"][e] == int(ep["completion"]) and fin["step_count"][e] == Te, tag
            assert fin["status"][e] & ~4 == 0, tag


def test_so_sfjsp_matches_reference_fixtures(torch_gpu):
    """SO_SFJSP (DDQN's environment): per-step and fused kernels vs the reference traces (SO_SFJSP.py:85-222)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_SO_SFJSP
    insts, eps, base = H.load_suite("so_sfjsp")
    s = H.instance_set_from([insts[ep["inst"]] for ep in eps])
    T = max(ep["T"] for ep in eps)
    actions = np.zeros((T, len(eps), 2), np.uint8)
    for e, ep in enumerate(eps):
        actions[:ep["T"], e] = ep["actions"]
    actions = torch.from_numpy(actions).cuda()
    for mode in ("step", "rollout"):
        b = EnvBatch(s, len(eps), variant=VARIANT_SO_SFJSP, rng_seed=base)
        assert b.state_size == 18
        st0 = b.reset().cpu().numpy()
        for e, ep in enumerate(eps):
            H.assert_state_close(st0[e], ep["state0"], "sf ep %d reset" % e, sf=True)
        if mode == "step":
            rewards = np.zeros((T, len(eps))); states = np.zeros((T, len(eps), 18))
            for t in range(T):
                st, r, d = b.step(actions[t])
                rewards[t] = r.cpu().numpy(); states[t] = st.cpu().numpy()
            trace = None
        else:
            trace, rw, st = b.rollout(actions)
            trace = trace.cpu().numpy(); rewards = rw.cpu().numpy(); last = st.cpu().numpy()
        fin = {k: v.cpu().numpy() for k, v in b.read().items()}
        tend = b.machine_time_end().cpu().numpy()
        for e, ep in enumerate(eps):
            Te = ep["T"]
            tag = "so_sfjsp %s episode %d (%s)" % (mode, e, insts[ep["inst"]].name)
            assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(ep["reward"])), tag + " reward"
            if trace is not None:
                assert np.array_equal(trace[:Te, e, 0], ep["k"]) and np.array_equal(trace[:Te, e, 1], ep["m"]), tag
                H.assert_state_close(last[e], ep["state_last"], tag, sf=True)
            else:
                H.assert_state_close(states[Te - 1, e], ep["state_last"], tag, sf=True)
                if "states" in ep:
                    H.assert_state_close(states[:Te, e], ep["states"], tag, sf=True)
            M = insts[ep["inst"]].M
            assert np.array_equal(tend[e, :M], ep["tend"]), tag
            assert fin["makespan"][e] == ep["final"][0] == fin["completion_time"][e], tag
            assert fin["delay_time_sum"][e] == ep["final"][1] and fin["step_count"][e] == Te, tag
            assert fin["status"][e] & ~4 == 0, tag


def test_mo_dfjsp_breakdown_matches_reference_fixtures(torch_gpu):
    """MO_DFJSP_breakdown (BASELINE config 5): order arrivals + machine breakdowns + energy, 12 x 10 rules,
    reward policies 0..3, against the reference traces (MO_DFJSP_breakdown.py:189-447)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_DFJSP
    insts, eps, base = H.load_suite("mo_dfjsp")
    s = H.instance_set_from([insts[ep["inst"]] for ep in eps])
    T = max(ep["T"] for ep in eps)
    actions = np.zeros((T, len(eps), 2), np.uint8)
    for e, ep in enumerate(eps):
        actions[:ep["T"], e] = ep["actions"]
    actions = torch.from_numpy(actions).cuda()
    mo = torch.from_numpy(np.stack([np.maximum(ep["mo"], 0.0) for ep in eps])).cuda()      # -1 = None -> unused
    for mode in ("step", "rollout"):
        b = EnvBatch(s, len(eps), variant=VARIANT_MO_DFJSP, rng_seed=base)
        assert b.state_size == 30
        st0 = b.reset().cpu().numpy()
        for e, ep in enumerate(eps):
            H.assert_state_close(st0[e], ep["state0"], "dyn ep %d reset" % e, dyn=True)
        if mode == "step":
            rewards = np.zeros((T, len(eps))); states = np.zeros((T, len(eps), 30))
            done_at = np.full(len(eps), -1)
            for t in range(T):
                alive = (b.done == 0).cpu().numpy()
                st, r, d = b.step(actions[t], mo=mo)
                rewards[t] = r.cpu().numpy(); states[t] = st.cpu().numpy()
                d = d.cpu().numpy()
                done_at[(done_at < 0) & alive & (d == 1)] = t
            trace = None
            for e, ep in enumerate(eps):
                assert done_at[e] == ep["T"] - 1, "dyn episode %d finished at step %d, reference %d" % (e, done_at[e], ep["T"] - 1)
        else:
            trace, rw, st = b.rollout(actions, mo=mo)
            trace = trace.cpu().numpy(); rewards = rw.cpu().numpy(); last = st.cpu().numpy()
        fin = {k: v.cpu().numpy() for k, v in b.read().items()}
        tend = b.machine_time_end().cpu().numpy()
        for e, ep in enumerate(eps):
            Te = ep["T"]
            tag = "mo_dfjsp %s episode %d (%s)" % (mode, e, insts[ep["inst"]].name)
            assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(ep["reward"])), tag + " reward"
            if trace is not None:
                assert np.array_equal(trace[:Te, e, 0], ep["k"]) and np.array_equal(trace[:Te, e, 1], ep["m"]), tag
                H.assert_state_close(last[e], ep["state_last"], tag, dyn=True)
            else:
                H.assert_state_close(states[Te - 1, e], ep["state_last"], tag, dyn=True)
                if "states" in ep:
                    H.assert_state_close(states[:Te, e], ep["states"], tag, dyn=True)
            M = insts[ep["inst"]].M
            assert np.array_equal(tend[e, :M], ep["tend"]), tag
            assert fin["makespan"][e] == ep["final"][0] and fin["delay_time_sum"][e] == ep["final"][1], tag
            assert fin["completion_time"][e] == int(ep["completion"]) and fin["step_count"][e] == Te, tag
            assert fin["energy_consumption"][e] == int(ep["energy"]), tag + " energy"
            assert fin["status"][e] & ~4 == 0, tag


def test_mo_dfjsp_single_env_mirror(torch_gpu, tmp_path):
    """environments.MO_DFJSP_breakdown.MO_DFJSP_Environment: the reference's constructor / step signature over
    a CSV folder (written from a fixture instance), including the 1-element flat action (:191-192)."""
    import csv
    from deep_reinforcement_learning_for_fjsp_amd.environments import MO_DFJSP_Environment
    from deep_reinforcement_learning_for_fjsp_amd.utilities.Utility_Class import MyError
    insts, eps, base = H.load_suite("mo_dfjsp")
    ep = next(e for e in eps if insts[e["inst"]].name.startswith("gen") and e["mo"][0] == 1)
    a = insts[ep["inst"]]
    folder = tmp_path / "D0"
    folder.mkdir()
    koff = np.concatenate(([0], np.cumsum(a.Jr)))
    with open(folder / "based_data.csv", "w", newline="") as f:
        csv.writer(f).writerows([["kind_count", "machine_count", "order_count", "DDT"], [a.R, a.M, a.S, a.ddt]])
    with open(folder / "process_data.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kind", "task", "machine_selectable", "process_time", "power"])
        for r in range(a.R):
            for j in range(int(a.Jr[r])):
                k = int(koff[r]) + j
                ms = tuple(int(m) for m in a.elig_list[k, :a.elig_n[k]])
                w.writerow([r, j, ms, tuple(int(a.p[k, m]) for m in ms), tuple(int(a.power[k, m]) for m in ms)])
    with open(folder / "order_data.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["order", "time_arrive", "time_delivery", "kind_number"])
        for so in range(a.S):
            w.writerow([so, int(a.arrive[so]), int(a.delivery[so]), tuple(int(c) for c in a.count[so])])
    with open(folder / "machine_data.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["machine", "idle_power", "breakdown_start", "breakdown_end"])
        off = np.concatenate(([0], np.cumsum(a.bk_n)))
        for m in range(a.M):
            if a.bk_n[m] == 0:
                w.writerow([m, int(a.idle_power[m])])
            for q in range(int(off[m]), int(off[m + 1])):
                w.writerow([m, int(a.idle_power[m]), int(a.bk[q, 0]), int(a.bk[q, 1])])
    env = MO_DFJSP_Environment(use_instance=False, path=str(tmp_path), file_name="D0", rng_seed=ep["rng_seed"])
    assert env.state_size == 30 and env.actions_size == [12, 10]
    st = env.reset()
    H.assert_state_close(st, ep["state0"], "mirror reset", dyn=True)
    t = 0
    while not env.done:
        a0, a1 = int(ep["actions"][t][0]), int(ep["actions"][t][1])
        act = [a0, a1] if t % 2 else [a0 * 10 + a1]          # both action forms
        st, r, d = env.step(act, reward_policy=1)
        assert r == ep["reward"][t] and env.step_time == ep["step_time"][t] and env.delay_time_sum == ep["delay"][t]
        t += 1
    assert t == ep["T"] and env.energy_consumption == int(ep["energy"]) and env.completion_time == int(ep["completion"])
    H.assert_state_close(st, ep["state_last"], "mirror last", dyn=True)
    assert max(mv.time_end for mv in env.machine_dict.values()) == ep["final"][0]
    env.reset()
    with pytest.raises(MyError):
        env.step([0, 0], reward_policy=7)
    with pytest.raises(MyError):
        env.step([12, 0], reward_policy=0)


def test_full_size_batch_against_oracle_and_invariants(torch_gpu):
    """BASELINE config 2 at full size: 4096 generated 10x5 instances (seeds 1000+i), random policy.

    Oracle comparison on a sample of envs (the oracle finishes those in seconds) plus
    size-independent properties on all 4096: one step per operation, rewards telescope to
    -delay_time_sum, the fused kernel and the per-step kernel agree bit for bit, and an
    autoreset pass reproduces the first episode (reset is idempotent)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
    N = 4096
    s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    K = np.array([s.dims(i)["K"] for i in range(N)])
    T = int(K.max())
    rs = np.random.RandomState(123)
    actions_h = np.stack([rs.randint(0, 6, (T, N)), rs.randint(0, 5, (T, N))], 2).astype(np.uint8)
    actions = torch.from_numpy(actions_h).cuda()
    b = EnvBatch(s, N, rng_seed=99)
    st0 = b.reset().clone()
    states, rewards, dones = [], [], []
    for t in range(T):
        st, r, d = b.step(actions[t])
        states.append(st.clone()); rewards.append(r.clone()); dones.append(d.clone())
    rewards = torch.stack(rewards).cpu().numpy(); dones = torch.stack(dones).cpu().numpy()
    states = torch.stack(states).cpu().numpy()
    fin = {k: v.cpu().numpy() for k, v in b.read().items()}
    assert (fin["done"] == 1).all() and np.array_equal(fin["step_count"], K)          # one step per operation
    assert np.array_equal(-rewards.sum(0).astype(np.int64), fin["delay_time_sum"])     # telescoping rewards
    first_done = dones.argmax(0)
    assert np.array_equal(first_done + 1, K)
    assert (fin["status"] & ~4 == 0).all()
    # fused kernel == per-step kernel on all 4096 envs
    b2 = EnvBatch(s, N, rng_seed=99)
    b2.reset()
    tr, rw, st_last = b2.rollout(actions)
    fin2 = {k: v.cpu().numpy() for k, v in b2.read().items()}
    assert np.array_equal(H.bits(rw.cpu().numpy()), H.bits(rewards))
    for key in ("delay_time_sum", "makespan", "step_time", "step_count", "completion_time"):
        assert np.array_equal(fin[key], fin2[key]), key
    last = states[K - 1, np.arange(N)]
    assert np.array_equal(H.bits(st_last.cpu().numpy()), H.bits(last))
    # oracle on a sample
    tr = tr.cpu().numpy()
    for e in list(range(0, N, 64)):
        a = s.arrays(e)
        want = H.play_oracle(a, a.x, actions_h[:, e], b.env_seed(e))
        Te = want["T"]
        assert Te == K[e]
        assert np.array_equal(tr[:Te, e, 0], want["k"]) and np.array_equal(tr[:Te, e, 1], want["m"]), e
        assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(want["reward"])), e
        H.assert_state_close(st0[e].cpu().numpy(), want["state0"], "env %d reset" % e)
        H.assert_state_close(states[:Te, e], want["states"], "env %d" % e)
        assert fin["makespan"][e] == want["makespan"] and fin["delay_time_sum"][e] == want["delay_time_sum"]
    # autoreset: a done env is reset inside the next step; reset state must equal the first one.
    # (the random.choice stream continues across resets, so only deterministic rules are replayed)
    det = np.stack([np.full((T, N), 2), np.full((T, N), 0)], 2).astype(np.uint8)
    b3 = EnvBatch(s, N, rng_seed=1)
    b3.reset()
    det_d = torch.from_numpy(det).cuda()
    ep1 = []
    for t in range(T):
        st, r, d = b3.step(det_d[t], autoreset=True)
        ep1.append((st.clone(), r.clone(), d.clone()))
    # envs with K == T finished exactly at the last step; step again with autoreset -> first step of episode 2
    st, r, d = b3.step(det_d[0], autoreset=True)
    idx = np.nonzero(K == T)[0]
    assert len(idx) > 0
    assert np.array_equal(H.bits(st.cpu().numpy()[idx]), H.bits(ep1[0][0].cpu().numpy()[idx]))
    assert np.array_equal(H.bits(r.cpu().numpy()[idx]), H.bits(ep1[0][1].cpu().numpy()[idx]))


def test_mo_dfjsp_full_size_batch_against_oracle_and_invariants(torch_gpu):
    """BASELINE config 5 at full size: 4096 MO_DFJSP_breakdown envs over the reference's industrial / HMPSAC
    instances (round-robin), random 12 x 10 rule policy, tardiness reward.  Oracle comparison on a sample of
    envs plus size-independent properties on all 4096: one step per operation of every order, rewards
    telescope to -delay_time_sum, energy >= the processing energy floor, every order-arrival LP was served."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_DFJSP
    insts, _, _ = H.load_suite("mo_dfjsp")
    insts = [a for a in insts if not a.name.startswith("gen")]
    s = H.instance_set_from(insts)
    N, n_inst = 4096, len(insts)
    ops = np.array([int((a.count.sum(0) * a.Jr).sum()) for a in insts])
    T = int(ops.max())
    rs = np.random.RandomState(321)
    actions_h = np.stack([rs.randint(0, 12, (T, N)), rs.randint(0, 10, (T, N))], 2).astype(np.uint8)
    actions = torch.from_numpy(actions_h).cuda()
    b = EnvBatch(s, N, variant=VARIANT_MO_DFJSP, rng_seed=2024)
    mo = torch.zeros(N, 4, dtype=torch.float64, device="cuda"); mo[:, 0] = 1.0
    b.reset()
    rsum = torch.zeros(N, dtype=torch.float64, device="cuda")
    sample = list(range(0, N, 173))
    rew_s = np.zeros((T, len(sample)))
    first_done = torch.full((N,), -1, dtype=torch.int64, device="cuda")
    for t in range(T):
        alive = b.done == 0
        st, r, d = b.step(actions[t], mo=mo)
        rsum += torch.where(alive, r, torch.zeros_like(r))
        first_done = torch.where(alive & (d == 1), torch.full_like(first_done, t), first_done)
        rew_s[t] = r[sample].cpu().numpy()
    fin = {k: v.cpu().numpy() for k, v in b.read().items()}
    K_env = ops[np.arange(N) % n_inst]
    assert (fin["done"] == 1).all() and np.array_equal(fin["step_count"], K_env)
    assert np.array_equal(first_done.cpu().numpy() + 1, K_env)
    assert np.array_equal(-rsum.cpu().numpy().astype(np.int64), fin["delay_time_sum"])       # telescoping rewards
    assert (fin["status"] & ~4 == 0).all()
    floor = np.array([int(np.where(a.p > 0, a.power.astype(np.int64) * a.p, 1 << 60).min(1) @ np.repeat(a.count.sum(0), a.Jr))
                      for a in insts])
    assert (fin["energy_consumption"] >= floor[np.arange(N) % n_inst]).all()
    assert (fin["completion_time"] <= fin["makespan"]).all()           # a breakdown at a task's end only delays the machine
    arrivals = np.array([a.S - 1 for a in insts])[np.arange(N) % n_inst].sum()
    assert 0 < b.lp_solves <= arrivals                                 # several arrivals inside one step share one LP
    for j, e in enumerate(sample):
        a = insts[e % n_inst]
        want = H.play_oracle(a, a.x, actions_h[:, e], b.env_seed(e), variant=4, mo=(1, 0, 0, 0))
        Te = want["T"]
        assert Te == K_env[e]
        assert np.array_equal(H.bits(rew_s[:Te, j]), H.bits(want["reward"])), e
        assert fin["makespan"][e] == want["makespan"] and fin["delay_time_sum"][e] == want["delay_time_sum"], e
        assert fin["energy_consumption"][e] == want["energy"] and fin["completion_time"][e] == want["completion_time"], e


@pytest.mark.parametrize("shape", ["small", "big", "jobs", "long", "rows"])
@pytest.mark.parametrize("variant", [0, 1, 2, 4, 5])
def test_randomised_differential_vs_oracle(torch_gpu, variant, shape):
    """Beyond the committed reference traces: freshly generated instances of mixed shape -- "small": 96 x (1-6
    kinds, 1-4 stages, 1-12 machines, 1-4 jobs per kind), also shops with more machines than operation types;
    "big": 24 x (8-24 kinds, 3-8 stages -> up to 192 operation types in one batch, i.e. the 2- and 4-chunk
    kernels, 8-32 machines, 1-3 jobs per kind); "jobs": 32 x (1-4 kinds with up to 60 jobs each); "rows": 50 x (1-15 kinds of ONE
    job, at most 64 operation types and 8 machines, one order: the shapes the 16-lane-row kernels of fjsp_group.hip take for
    SO_FJSSP / SO_DFJSP / MO_FJSSP_discretes -- operation counts on both sides of 16 / 32 / 48, a last wave that is not full,
    clocks beyond 2^16) -- 1-3 orders where the variant has arrivals, dense breakdown
    windows for the dynamic variant, random actions over the variant's whole action space, HIP kernels vs the C
    oracle (which is pinned to the reference on 2 921 episodes): choices, rewards, clocks and totals bit for bit,
    states up to the pow() entries."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
    big = shape == "big"
    NI = {"small": 96, "big": 24, "jobs": 32, "long": 48, "rows": 50}[shape]           # instances; they sit at an offset inside the set
    OFF = 2
    N = NI + NI // 4                                           # environments: the last quarter shares instances
    fuzz = int(os.environ.get("FJSP_FUZZ_SEED", "0"))          # tools/fuzz_parity.sh sweeps this
    rs = np.random.RandomState(1000 + variant + {"small": 0, "big": 50, "jobs": 70, "long": 90, "rows": 110}[shape] + 1000 * fuzz)
    s = fi.InstanceSet(OFF + NI)
    multi = variant in (0, 4, 5) and shape != "rows"
    for i in range(OFF, OFF + NI):
        pmin, pmax = 1, int(rs.randint(2, 60))
        if shape == "long":      # one job per kind, processing times near the u16 limit: clocks beyond 2^22
            R = int(rs.randint(20, 41)); Jlo = int(rs.randint(3, 6)); M = int(rs.randint(2, 5)); nmax = 1
            pmin, pmax = 30000, 65535
        elif shape == "rows":      # one job per kind, <= 15 kinds, <= 64 operation types, <= 8 machines
            R = int(rs.randint(1, 16)); Jlo = int(rs.randint(1, max(2, min(6, 64 // R)))); M = int(rs.randint(1, 9)); nmax = 1
            if rs.rand() < 0.25:
                pmin, pmax = 2000, 65535
        elif shape == "jobs":      # few kinds, many jobs per kind (list positions, FIFO order, several job-table chunks)
            R = int(rs.randint(1, 5)); Jlo = int(rs.randint(2, 5)); M = int(rs.randint(2, 9)); nmax = int(rs.randint(20, 61))
        elif big:
            R = int(rs.randint(8, 25)); Jlo = int(rs.randint(3, 8)); M = int(rs.randint(8, 33)); nmax = int(rs.randint(1, 4))
        else:
            R = int(rs.randint(1, 7)); Jlo = int(rs.randint(1, 4)); M = int(rs.randint(1, 13)); nmax = int(rs.randint(1, 5))
        prm = fi.GenParams(R_min=R, R_max=R, J_min=Jlo, J_max=Jlo + int(rs.randint(0, 2)), M=M, p_min=pmin, p_max=pmax,
                           N_min=1, N_max=nmax, S=int(rs.randint(1, 4)) if multi else 1,
                           DDT=float(rs.choice([0.5, 1.0, 1.5])), t_si_min=20.0, t_si_max=80.0)
        seed = 50000 * (variant + 1) + i + {"small": 0, "big": 25000, "jobs": 12000, "long": 37000, "rows": 44000}[shape] + 1000003 * fuzz
        s.generate(i, seed, prm)
        while variant in (4, 5) and not (s.arrays(i).p > 0).any(axis=0).all():  # the reference divides by zero there
            seed += 7919
            s.generate(i, seed, prm)
        if variant == 4:
            s.generate_machine_data(i, seed, max_windows=4, window_gap=(1, 60), window_len=(1, 30))
    s.solve_fluid(OFF, NI)
    inst_arrs = [s.arrays(OFF + i) for i in range(NI)]
    arrs = [inst_arrs[e % NI] for e in range(N)]              # env e plays instance first + e % n_inst
    T = max(int((a.count.sum(0) * a.Jr).sum()) for a in arrs)
    n0, n1 = {0: (6, 5), 1: (20, 1), 2: (18, 1), 4: (12, 10), 5: (6, 5)}[variant]
    actions_h = np.stack([rs.randint(0, n0, (T, N)), rs.randint(0, n1, (T, N))], 2).astype(np.uint8)
    actions = torch.from_numpy(actions_h).cuda()
    b = EnvBatch(s, N, first=OFF, n_inst=NI, variant=variant, rng_seed=777 + variant)
    # step()'s extra arguments differ per environment: weight vectors with or without normalisers for the MO
    # variant (MO_FJSSP_discretes.py:232-244), reward policies 0..3 for the dynamic one (MO_DFJSP_breakdown.py:430-447)
    if variant == 2:
        kinds = rs.randint(0, 3, N)
        w0 = np.round(rs.rand(N), 2)
        mo_rows = [(1.0, 0.0, -1.0, -1.0) if k == 0 else (0.0, 1.0, -1.0, -1.0) if k == 1 else
                   (float(w0[e]), float(1.0 - w0[e]), float(rs.randint(20, 90)), float(rs.randint(30, 400))) for e, k in enumerate(kinds)]
    elif variant == 4:
        mo_rows = [(float(rs.randint(0, 4)), float(rs.randint(20, 90)), float(rs.choice([0.0, 17.0, 250.0])), float(rs.randint(500, 5000)))
                   for _ in range(N)]
    else:
        mo_rows = None
    mo = None if mo_rows is None else torch.tensor(mo_rows, dtype=torch.float64).cuda()
    if shape == "rows":      # (SO_FJSSP, SO_DFJSP, MO_FJSSP_discretes: the row kernels; the other variants keep the wave kernels)
        forced_wave = os.environ.get("FJSP_STEP_IMPL") == "wave"       # (tools/fuzz_parity.sh sweeps with the row kernels switched off too)
        assert b.kernel_family == (1 if variant in (0, 2, 5) and not forced_wave else 0)
    st0 = b.reset().cpu().numpy()
    S = b.state_size
    rewards = np.zeros((T, N)); states = np.zeros((T, N, S))
    for t in range(T):
        st, r, d = b.step(actions[t], mo=mo)
        rewards[t] = r.cpu().numpy(); states[t] = st.cpu().numpy()
    fin = {k: v.cpu().numpy() for k, v in b.read().items()}
    assert (fin["done"] == 1).all() and (fin["status"] & ~4 == 0).all()
    kw = {0: {}, 1: dict(sf=True), 2: dict(mo=True), 4: dict(dyn=True), 5: {}}[variant]
    for e, a in enumerate(arrs):
        want = H.play_oracle(a, a.x, actions_h[:, e], b.env_seed(e), variant=variant, mo=None if mo_rows is None else mo_rows[e])
        Te = want["T"]
        tag = "variant %d env %d (R=%d M=%d S=%d)" % (variant, e, a.R, a.M, a.S)
        assert fin["step_count"][e] == Te, tag
        assert np.array_equal(H.bits(rewards[:Te, e]), H.bits(want["reward"])), tag + " reward"
        H.assert_state_close(st0[e], want["state0"], tag + " reset", **kw)
        H.assert_state_close(states[:Te, e], want["states"], tag, **kw)
        assert fin["makespan"][e] == want["makespan"] and fin["delay_time_sum"][e] == want["delay_time_sum"], tag
        assert fin["step_time"][e] == want["step_time"][-1], tag
        if variant == 4:
            assert fin["energy_consumption"][e] == want["energy"] and fin["completion_time"][e] == want["completion_time"], tag
    # the fused T-step kernel (or its step-launch fallback) agrees with the per-step launches bit for bit
    b2 = EnvBatch(s, N, first=OFF, n_inst=NI, variant=variant, rng_seed=777 + variant)
    b2.reset()
    tr, rw, st_last = b2.rollout(actions, mo=mo)
    fin2 = {k: v.cpu().numpy() for k, v in b2.read().items()}
    assert np.array_equal(H.bits(rw.cpu().numpy()), H.bits(rewards))
    for key in ("delay_time_sum", "makespan", "step_time", "step_count", "completion_time"):
        assert np.array_equal(fin[key], fin2[key]), key
    Ks = fin["step_count"]
    assert np.array_equal(H.bits(st_last.cpu().numpy()), H.bits(states[Ks - 1, np.arange(N)]))
    # ... and so does the stateless form (no observation inside the kernel): same choices, rewards and totals
    b3 = EnvBatch(s, N, first=OFF, n_inst=NI, variant=variant, rng_seed=777 + variant)
    b3.reset()
    tr3, rw3, none = b3.rollout(actions, mo=mo, state=False)
    fin3 = {k: v.cpu().numpy() for k, v in b3.read().items()}
    assert none is None and torch.equal(tr3, tr) and np.array_equal(H.bits(rw3.cpu().numpy()), H.bits(rewards))
    for key in ("delay_time_sum", "makespan", "step_time", "step_count", "completion_time"):
        assert np.array_equal(fin[key], fin3[key]), key
    # autoreset: every env of `b` is done; a step with autoreset starts a fresh episode inside the launch (reset
    # observation taken from the per-instance cache) and must equal reset() + step() on a fresh batch.  A
    # deterministic rule pair is used (the random.choice stream continues across resets).
    det = torch.tensor({0: (2, 0), 1: (0, 0), 2: (0, 0), 4: (2, 0), 5: (2, 0)}[variant], dtype=torch.uint8).repeat(N, 1).cuda()
    fresh = EnvBatch(s, N, first=OFF, n_inst=NI, variant=variant, rng_seed=5)
    fresh.reset()
    st_f, r_f, d_f = fresh.step(det, mo=mo)
    st_a, r_a, d_a = b.step(det, autoreset=True, mo=mo)
    assert torch.equal(st_a, st_f) and torch.equal(r_a, r_f) and torch.equal(d_a, d_f)
    fa, ff = b.read(), fresh.read()
    for key in ("step_time", "step_count", "delay_time_sum", "completion_time", "makespan"):
        assert torch.equal(fa[key], ff[key]), key
    assert int((fa["status"] != 0).sum()) == 0                 # the restart also cleared the sticky status bits
    # masked reset one step into the episode: masked envs are back at their reset state and replay their first
    # step, the others are untouched
    mask_h = np.arange(N) % 3 == 0
    mask = torch.from_numpy(mask_h.astype(np.uint8)).cuda()
    first_step = st_f.clone()
    st_m = fresh.reset(mask).clone()
    assert np.array_equal(H.bits(st_m.cpu().numpy()[mask_h]), H.bits(st0[mask_h]))
    assert torch.equal(st_m[~mask.bool()], first_step[~mask.bool()])
    st_2, r_2, _ = fresh.step(det, mo=mo)
    assert torch.equal(st_2[mask.bool()], first_step[mask.bool()]) and torch.equal(r_2[mask.bool()], r_f[mask.bool()])


def test_instance_sharing_and_masked_reset(torch_gpu):
    """env e plays instance e % n_inst; reset(mask) restarts only the masked envs (the others keep going)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
    s = fi.InstanceSet(3).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    N = 10                                          # not a multiple of the 4 envs per workgroup
    b = EnvBatch(s, N, rng_seed=0)
    act = torch.tensor([[2, 0]] * N, dtype=torch.uint8, device="cuda")
    first = b.reset().clone()
    for e in range(N):
        assert torch.equal(first[e], first[e % 3])
    hist = []
    for t in range(12):
        st, r, d = b.step(act)
        hist.append((st.clone(), r.clone()))
        for e in range(N):                          # deterministic rules: envs sharing an instance move in lock-step
            assert torch.equal(st[e], st[e % 3]) and r[e] == r[e % 3]
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda")
    mask[[1, 4, 9]] = 1
    before = b.state.clone()
    st = b.reset(mask)
    for e in range(N):
        if mask[e]:
            assert torch.equal(st[e], first[e])
        else:
            assert torch.equal(st[e], before[e])
    for t in range(12):
        st, r, d = b.step(act)
        for e in (1, 4, 9):                         # the restarted envs replay their first episode
            assert torch.equal(st[e], hist[t][0][e]) and r[e] == hist[t][1][e]
    rd = b.read()
    assert rd["step_count"].cpu().tolist() == [12 if m else 24 for m in mask.cpu().tolist()]
    assert int(rd["status"].abs().sum()) == 0


def test_error_behaviour(torch_gpu):
    """MyError on undefined rules, error on stepping a finished episode, picklable env."""
    import pickle
    from deep_reinforcement_learning_for_fjsp_amd.environments import SO_FJSSP_Environment
    from deep_reinforcement_learning_for_fjsp_amd.utilities.Utility_Class import MyError
    env = SO_FJSSP_Environment(use_instance=True, DDT=1.0, M=6, S=1, seed=11)
    s0 = env.reset()
    assert s0.shape == (20,) and s0.dtype == np.float64 and s0[0] == 6.0
    with pytest.raises(MyError):
        env.step([6, 0])
    with pytest.raises(MyError):
        env.step([0, 5])
    total = 0
    while not env.done:
        s, r, d = env.step([2, 0])
        assert isinstance(r, int)
        total += r
    assert -total == env.delay_time_sum and env.reward_sum == total
    assert max(v.time_end for v in env.machine_dict.values()) >= env.step_time > 0
    with pytest.raises(ValueError):
        env.step([2, 0])
    env2 = pickle.loads(pickle.dumps(env))
    s1 = env2.reset()
    assert np.array_equal(H.bits(s1), H.bits(s0))
    # the MO mirror: same protocol with step()'s extra arguments
    from deep_reinforcement_learning_for_fjsp_amd.environments import MO_FJSSP_Environment
    mo = MO_FJSSP_Environment(use_instance=True, DDT=1.0, M=6, S=1, seed=11)
    st = mo.reset()
    assert st.shape == (25,) and st[1] == 6.0
    with pytest.raises(IndexError):
        mo.step(18, weight_vector=(0, 1))
    with pytest.raises(MyError):
        mo.step(3, weight_vector=(0.5, 0.5))
    tot = 0
    while not mo.done:
        st, r, d = mo.step(3, weight_vector=(1, 0))
        tot += r
    assert -tot == mo.completion_time == max(v.time_end for v in mo.machine_dict.values())


def test_shards_concatenate_to_the_unsharded_batch(torch_gpu):
    """results are identical for any number of GPUs.  Two half batches built the way two ranks
    build them (instances, action tensor and random.choice streams from the GLOBAL env id) replay, env for env
    and bit for bit, what the unsharded batch does -- the random rules (task rule 6, machine rule 5) included."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, global_actions
    N, T = 96, 52
    whole = fi.InstanceSet(N).generate_range(7000, fi.bench_10x5_params()).solve_fluid()
    acts = global_actions(11, 0, N, T, 6, 5)
    acts[:, ::3] = (5, 4)                            # a third of the envs play random.choice / random.choice only
    def play(insts, n, first_env, a):
        b = EnvBatch(insts, n, rng_seed=77, first_env=first_env)
        out = [b.reset().clone()]
        tr = torch.zeros(n, 2, dtype=torch.int16, device=b.device)
        kms, rws = [], []
        a = torch.from_numpy(np.ascontiguousarray(a)).to(b.device)
        for t in range(T):
            s, r, _ = b.step(a[t], autoreset=True, trace_out=tr)
            out.append(s.clone()); kms.append(tr.clone()); rws.append(r.clone())
        fin = b.read()
        return (torch.stack(out).cpu().numpy(), torch.stack(kms).cpu().numpy(), torch.stack(rws).cpu().numpy(),
                fin["delay_time_sum"].cpu().numpy(), fin["makespan"].cpu().numpy())
    ref = play(whole, N, 0, acts)
    assert (ref[1] >= 0).all()
    for lo, hi in ((0, 40), (40, 96)):               # uneven shards, the second one not starting at a multiple of 4 x 8
        part = fi.InstanceSet(hi - lo).generate_range(7000 + lo, fi.bench_10x5_params()).solve_fluid()
        a = global_actions(11, lo, hi - lo, T, 6, 5)
        a[:, (-lo) % 3::3] = (5, 4)
        assert np.array_equal(a, acts[:, lo:hi])
        got = play(part, hi - lo, lo, a)
        assert np.array_equal(H.bits(got[0]), H.bits(ref[0][:, lo:hi])), "states of shard [%d, %d)" % (lo, hi)
        assert np.array_equal(got[1], ref[1][:, lo:hi]), "chosen (k, m) of shard [%d, %d)" % (lo, hi)
        assert np.array_equal(H.bits(got[2]), H.bits(ref[2][:, lo:hi]))
        assert np.array_equal(got[3], ref[3][lo:hi]) and np.array_equal(got[4], ref[4][lo:hi])


def test_config3_per_gpu_workload_in_eight_shards(torch_gpu):
    """BASELINE configs[3]: 32 768 envs sharded 8-way.  The per-GPU workload of that configuration on ONE GPU: eight
    shards of 4 096 envs built the way eight ranks build them (generator seeds, action tensor and random.choice streams from
    the GLOBAL env id), 24 autoreset steps each.  Checked: (a) one shard against the unsharded 32 768-env batch, bit for bit
    (states, rewards, chosen pairs) -- the unsharded batch runs the large-batch variant of the row kernels, the shard the
    small-batch one, so this is also their equality at full size; (b) an oracle replay of a sample of envs across all
    shards; (c) size-independent properties on all 32 768 envs: no error status, one dispatch per step, rewards telescope
    to minus the tardiness of the finished episodes."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, global_actions
    N, W, T = 32768, 8, 24
    per = N // W
    prm = fi.bench_10x5_params()

    def play(n, first_env, keep_states):
        insts = fi.InstanceSet(n).generate_range(1000 + first_env, prm).solve_fluid()
        b = EnvBatch(insts, n, rng_seed=20260, first_env=first_env)
        a = torch.from_numpy(global_actions(4242, first_env, n, T, 6, 5)).to(b.device)
        st0 = b.reset().clone()
        tr = torch.zeros(n, 2, dtype=torch.int16, device=b.device)
        rws, kms, sts = [], [], []
        for t in range(T):
            s, r, _ = b.step(a[t], autoreset=True, trace_out=tr)
            rws.append(r.clone()); kms.append(tr.clone())
            if keep_states:
                sts.append(s.clone())
        fin = b.read()
        assert int((fin["status"] != 0).sum()) == 0
        return insts, b, st0, torch.stack(rws), torch.stack(kms), (torch.stack(sts) if keep_states else None), fin

    _, whole_b, st0_w, rw_w, km_w, _, fin_w = play(N, 0, False)
    assert whole_b.kernel_family == 1
    assert bool((km_w >= 0).all())                                     # every env dispatched one operation per step
    rs = np.random.RandomState(5)
    for g in range(W):
        lo = g * per
        insts, b, st0, rw, km, _, fin = play(per, lo, False)
        assert torch.equal(st0, st0_w[lo:lo + per]) and torch.equal(rw, rw_w[:, lo:lo + per]) and torch.equal(km, km_w[:, lo:lo + per]), g
        for key in ("delay_time_sum", "makespan", "step_time", "step_count"):
            assert torch.equal(fin[key], fin_w[key][lo:lo + per]), (g, key)
        # oracle replay of two envs of this shard: its first T steps (no episode of the 10x5 workload is shorter than 30)
        acts_h = global_actions(4242, lo, per, T, 6, 5)
        for e in rs.randint(0, per, 2):
            a = insts.arrays(int(e))
            want = H.play_oracle(a, a.x, np.concatenate([acts_h[:, e], np.zeros((64, 2), np.uint8)]), b.env_seed(int(e)))
            assert want["T"] >= T
            assert np.array_equal(km[:, e, 0].cpu().numpy(), want["k"][:T]) and np.array_equal(km[:, e, 1].cpu().numpy(), want["m"][:T])
            assert np.array_equal(H.bits(rw[:, e].cpu().numpy()), H.bits(want["reward"][:T]))
    # rewards telescope: minus their sum over the T steps is the tardiness accumulated so far (no episode has ended yet)
    assert torch.equal((-rw_w.sum(0)).to(torch.int64), fin_w["delay_time_sum"])


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_steps_without_a_state_do_not_change_later_states(torch_gpu, variant):
    """fjsp_env_step with d_state == NULL skips the observation; the next step that returns a state rebuilds
    v(t-1) first, so every returned state equals the one an always-observing run returns (bit for bit)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, global_actions
    N, T = 70, 44                                    # (not a multiple of 4: the last workgroup is partial)
    insts = fi.InstanceSet(N).generate_range(8100, fi.bench_10x5_params()).solve_fluid()
    na = {0: (6, 5), 1: (20, 1), 2: (18, 1)}[variant]
    acts = torch.from_numpy(global_actions(3, 0, N, T, *na)).cuda()
    mo = None
    if variant == 2:
        mo = torch.tensor([[0.5, 0.5, 100.0, 1000.0]], dtype=torch.float64).repeat(N, 1).cuda()
    a = EnvBatch(insts, N, variant=variant, rng_seed=5)
    b = EnvBatch(insts, N, variant=variant, rng_seed=5)
    a.reset(); b.reset()
    pattern = [True, False, False, True, True, False, True, False, False, False, True]
    for t in range(T):
        sa, ra, da = a.step(acts[t], autoreset=True, mo=mo)
        with_state = pattern[t % len(pattern)]
        sb, rb, db = b.step(acts[t], autoreset=True, mo=mo, state=with_state)
        assert torch.equal(ra, rb) and torch.equal(da, db), "step %d" % t
        if with_state:
            assert np.array_equal(H.bits(sa.cpu().numpy()), H.bits(sb.cpu().numpy())), "state after step %d" % t
        else:
            assert sb is None
    fa, fb = a.read(), b.read()
    for k in fa:
        assert torch.equal(fa[k], fb[k]), k
    # the fused kernel without a final state, then a per-step call that wants one
    c = EnvBatch(insts, N, variant=variant, rng_seed=5); c.reset()
    d = EnvBatch(insts, N, variant=variant, rng_seed=5); d.reset()
    c.rollout(acts[:10], mo=mo, state=False)
    for t in range(10):
        d.step(acts[t], mo=mo)
    sc, _, _ = c.step(acts[10], mo=mo)
    sd, _, _ = d.step(acts[10], mo=mo)
    assert np.array_equal(H.bits(sc.cpu().numpy()), H.bits(sd.cpu().numpy()))


def test_wrong_tensor_arguments_are_rejected_before_any_launch(torch_gpu):
    """The C ABI takes raw pointers; EnvBatch checks every tensor's extent, type and device first."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
    N = 8
    b = EnvBatch(fi.InstanceSet(N).generate_range(1, fi.bench_10x5_params()).solve_fluid(), N, variant=2)
    b.reset()
    good = torch.zeros(N, 2, dtype=torch.uint8, device=b.device)
    with pytest.raises(ValueError):
        b.step(torch.zeros(N, dtype=torch.uint8, device=b.device))              # flat [N] actions
    with pytest.raises(ValueError):
        b.step(torch.zeros(N + 1, 2, dtype=torch.uint8, device=b.device))
    with pytest.raises(ValueError):
        b.step(good, mo=torch.zeros(N, 3, dtype=torch.float64, device=b.device))
    with pytest.raises(ValueError):
        b.step(good, state_out=torch.zeros(N, b.state_size, dtype=torch.float32, device=b.device))
    with pytest.raises(ValueError):
        b.step(good, reward_out=torch.zeros(N, dtype=torch.float64))             # host tensor as an output
    with pytest.raises(ValueError):
        b.step(good, done_out=torch.zeros(N - 1, dtype=torch.uint8, device=b.device))
    with pytest.raises(ValueError):
        b.reset(mask=torch.ones(N - 2, dtype=torch.uint8, device=b.device))
    with pytest.raises(ValueError):
        b.reset(out=torch.zeros(N, b.state_size + 1, dtype=torch.float64, device=b.device))
    with pytest.raises(ValueError):
        b.rollout(torch.zeros(5, N, dtype=torch.uint8, device=b.device))
    # inputs of another dtype / on the host are converted, as before
    b.step(torch.zeros(N, 2, dtype=torch.int64), mo=torch.tensor([[0.0, 1.0, 0.0, 0.0]] * N, dtype=torch.float32))
    assert int((b.read()["status"] != 0).sum()) == 0
    # an action tensor at an odd address (the kernels read a pair as one 16-bit word): the C ABI refuses it, EnvBatch
    # re-homes it and the step is the one the aligned tensor gives
    from deep_reinforcement_learning_for_fjsp_amd import _capi
    odd = torch.zeros(2 * N + 1, dtype=torch.uint8, device=b.device)[1:].view(N, 2)
    assert odd.data_ptr() & 1 and odd.is_contiguous()
    rc = b._lib.fjsp_env_step(b._h, _capi.C.c_void_p(odd.data_ptr()), None, 0, None, None, None, None)
    assert rc == -1 and b"2-byte aligned" in b._lib.fjsp_last_error()
    odd[:] = 1
    c = EnvBatch(fi.InstanceSet(N).generate_range(1, fi.bench_10x5_params()).solve_fluid(), N, variant=2)
    c.reset(); b.reset()
    s1, r1, _ = b.step(odd, mo=torch.tensor([[0.0, 1.0, 0.0, 0.0]] * N, dtype=torch.float64))
    s2, r2, _ = c.step(torch.ones(N, 2, dtype=torch.uint8, device=b.device), mo=torch.tensor([[0.0, 1.0, 0.0, 0.0]] * N, dtype=torch.float64))
    assert torch.equal(s1, s2) and torch.equal(r1, r2)


def test_async_arrival_service_keeps_every_env_on_its_own_trajectory(torch_gpu):
    """fjsp_env_step_async: envs that reach an order arrival park while the LP is solved in the background and the
    rest of the batch keeps stepping.  Every env must still walk exactly the trajectory the blocking fjsp_env_step
    gives it for the same action sequence: same rewards step by step, same final makespan / tardiness / energy."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_DFJSP, global_actions
    insts, _, _ = H.load_suite("mo_dfjsp")
    s = H.instance_set_from(insts)
    N = 96
    acts = torch.from_numpy(global_actions(21, 0, N, 1400, 12, 10)).cuda()      # [T, N, 2]
    mo = torch.zeros(N, 4, dtype=torch.float64, device="cuda"); mo[:, 0] = 1.0
    idx = torch.arange(N, device="cuda")
    # ---- blocking service
    a = EnvBatch(s, N, variant=VARIANT_MO_DFJSP, rng_seed=5)
    a.reset()
    rew_a = torch.zeros(1400, N, dtype=torch.float64, device="cuda")
    steps_a = torch.zeros(N, dtype=torch.int64, device="cuda")
    t = 0
    while True:
        live = a.done == 0
        _, r, d = a.step(acts[t], mo=mo)
        rew_a[t] = torch.where(live, r, torch.zeros_like(r))
        steps_a += live.long()
        t += 1
        if t % 50 == 0 and bool((a.done != 0).all()):
            break
        assert t < 1400
    fa = a.read()
    # ---- asynchronous service: per-env cursor into the same action sequences
    b = EnvBatch(s, N, variant=VARIANT_MO_DFJSP, rng_seed=5)
    b.reset()
    cursor = torch.zeros(N, dtype=torch.int64, device="cuda")
    rew_b = torch.zeros(1400, N, dtype=torch.float64, device="cuda")
    done_prev = torch.zeros(N, dtype=torch.bool, device="cuda")
    calls, parked_seen = 0, 0
    parked_prev = torch.zeros(N, dtype=torch.bool, device="cuda")
    junk_shown = 0
    while True:
        cur = acts[cursor.clamp(max=1399), idx]
        # the step of a parked env applies the action of the call in which it parked; while it stays parked and in the
        # call where it resumes its entry is not looked at (include/fjsp_amd.h): show it a DIFFERENT action there
        junk = torch.stack([(cur[:, 0] + 5) % 12, (cur[:, 1] + 3) % 10], 1).to(torch.uint8)
        shown = torch.where(parked_prev[:, None], junk, cur)
        junk_shown += int(parked_prev.sum()) if calls % 25 == 0 else 0
        _, r, d, ready = b.step_async(shown, mo=mo)
        parked_prev = ready == 0
        took = (ready != 0) & ~done_prev
        rew_b[cursor.clamp(max=1399), idx] = torch.where(took, r, rew_b[cursor.clamp(max=1399), idx])
        cursor += took.long()
        done_prev = done_prev | ((ready != 0) & (d != 0))
        calls += 1
        parked_seen = max(parked_seen, b.parked)          # (a host-side counter: no device round trip)
        if calls % 25 == 0:
            if bool(done_prev.all()):
                break
        assert calls < 6000
    b.flush_arrivals(mo)
    assert b.parked == 0 and parked_seen > 0
    fb = b.read()
    assert torch.equal(cursor, steps_a), "steps per env"
    assert torch.equal(rew_a, rew_b), "per-step rewards"
    for k in ("delay_time_sum", "makespan", "completion_time", "step_count", "energy_consumption", "done"):
        assert torch.equal(fa[k], fb[k]), k
    assert int(((fb["status"] & ~4) != 0).sum()) == 0
    assert b.lp_solves == a.lp_solves
    # the blocking entry points refuse to run while envs are parked
    c = EnvBatch(s, N, variant=VARIANT_MO_DFJSP, rng_seed=5); c.reset()
    from deep_reinforcement_learning_for_fjsp_amd._capi import FjspError
    refused = False
    for t in range(400):
        c.step_async(acts[t], mo=mo)
        try:
            c.step(acts[t], mo=mo)
        except FjspError as err:
            refused = err.code == -7          # FJSP_E_STATE
            break
    assert refused
    c.flush_arrivals(mo)
    c.step(acts[0], mo=mo)


def test_device_lp_equals_the_host_lp(torch_gpu):
    """csrc/fjsp_lp_device.hip restates the host simplex (csrc/fjsp_lp.cpp) pivot for pivot: on every instance of the
    mo_dfjsp / multiorder suites whose tableau fits the LDS, for the reset-time LP and for random live states (jobs spread
    over the stages, so precedence rows come and go with n_now == 0), the device's x equals the host's BIT FOR BIT."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_DFJSP
    rs = np.random.RandomState(17)
    checked = 0
    for suite, variant in (("mo_dfjsp", VARIANT_MO_DFJSP), ("multiorder", 0)):
        insts, _, _ = H.load_suite(suite)
        for i, a in enumerate(insts):
            s = H.instance_set_from([a])
            with H.env_var("FJSP_LP_IMPL", "device"):      # (small batches default to the host service)
                b = EnvBatch(s, 4, variant=variant, rng_seed=1)
            if not b.lp_on_device:
                continue                                   # (tableau beyond the LDS: this batch keeps the host service)
            koff = np.concatenate([[0], np.cumsum(a.Jr)])
            K, M = a.p.shape
            for trial in range(6):
                Q = np.zeros(K, np.int32); now = np.zeros(K, np.int32)
                for r in range(len(a.Jr)):
                    n = int(rs.randint(1, 25))
                    if trial == 0:
                        stages = np.zeros(n, np.int64)                             # every job at stage 0: the reset-time LP
                    else:
                        stages = rs.randint(0, a.Jr[r], n)                          # jobs spread over the stages (one stays at the last)
                        stages[0] = a.Jr[r] - 1 if trial % 2 else stages[0]
                    for j in range(a.Jr[r]):
                        Q[koff[r] + j] = max(1, int((stages <= j).sum()))           # tasks of (r, j) still unprocessed (class_FJSSP.py:234-235)
                        now[koff[r] + j] = int((stages == j).sum())                 # jobs waiting at (r, j)                 (:236-237)
                want, _ = fi.fluid_lp(a.Jr, a.p, Q, now)
                got = b.lp_device_solve(trial % 4, Q, now)[:K * M].reshape(K, M)
                assert np.array_equal(H.bits(got), H.bits(want)), "%s instance %d (%s) trial %d" % (suite, i, a.name, trial)
                checked += 1
    assert checked >= 12


def test_device_lp_service_leaves_every_trajectory_unchanged(torch_gpu):
    """The dynamic environment with its order-arrival LPs on the device (FJSP_LP_IMPL=device) against the same batch with
    FJSP_LP_IMPL=host:
    rewards step by step, final makespan / tardiness / energy and the number of LPs are identical."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_DFJSP, global_actions
    insts, _, _ = H.load_suite("mo_dfjsp")

    def tableau_bytes(a):        # rows x columns of the largest tableau of the instance (csrc/fjsp_lp_device.hip lp_device_lds_bytes)
        K, M = a.p.shape
        nr = K + M + (K - len(a.Jr))
        return nr * (int((a.p > 0).sum()) + 1 + nr + 1) * 8
    insts = [a for a in insts if tableau_bytes(a) < 130 * 1024]      # (the industrial folders and the generated ones; not data/HMPSAC)
    assert len(insts) >= 4
    s = H.instance_set_from(insts)
    N, T = 96, 1600
    acts = torch.from_numpy(global_actions(29, 0, N, T, 12, 10)).cuda()
    mo = torch.zeros(N, 4, dtype=torch.float64, device="cuda"); mo[:, 0] = 1.0

    def play(impl):
        with H.env_var("FJSP_LP_IMPL", impl):
            b = EnvBatch(s, N, variant=VARIANT_MO_DFJSP, rng_seed=5)
        b.reset()
        rew = torch.zeros(T, N, dtype=torch.float64, device="cuda")
        for t in range(T):
            live = b.done == 0
            _, r, d = b.step(acts[t], mo=mo)
            rew[t] = torch.where(live, r, torch.zeros_like(r))
            if t % 50 == 49 and bool((b.done != 0).all()):
                break
        return b, rew, b.read()

    dev, rew_d, fin_d = play("device")
    host, rew_h, fin_h = play("host")
    assert dev.lp_on_device == 1 and host.lp_on_device == 0
    assert dev.lp_device_pivots > 0 and host.lp_device_pivots == 0
    assert bool((fin_d["done"] != 0).all())
    assert torch.equal(rew_d, rew_h)
    for k in ("delay_time_sum", "makespan", "completion_time", "step_count", "energy_consumption", "done", "status"):
        assert torch.equal(fin_d[k], fin_h[k]), k
    assert dev.lp_solves == host.lp_solves > 0
