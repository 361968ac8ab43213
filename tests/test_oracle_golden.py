"""CPU: the C oracle reproduces every stored reference trace bit-exactly.

The fixtures were produced by tests/golden/make_golden.py from the reference
itself (see tests/golden/GENERATION_REPORT.txt); here the oracle replays the
stored actions on the stored instance arrays + fluid solution.
"""
import numpy as np
import pytest

from tests import helpers as H


@pytest.mark.parametrize("suite", H.SUITES + H.SOD_SUITES + H.MO_SUITES + H.SF_SUITES + H.DYN_SUITES + H.ORACLE_ONLY_SUITES)
def test_oracle_matches_reference_fixtures(built, suite):
    insts, eps, _ = H.load_suite(suite)
    assert eps, "empty fixture"
    for e, ep in enumerate(eps):
        a = insts[ep["inst"]]
        if suite in H.MO_SUITES:
            got = H.play_oracle(a, a.x, ep["actions"], ep["rng_seed"], variant=2, mo=ep["mo"])
            assert got["completion_time"] == int(ep["completion"])
        elif suite in H.DYN_SUITES:
            got = H.play_oracle(a, a.x, ep["actions"], ep["rng_seed"], variant=4, mo=ep["mo"])
            assert got["completion_time"] == int(ep["completion"]) and got["energy"] == int(ep["energy"])
        elif suite in H.SF_SUITES:
            got = H.play_oracle(a, a.x, ep["actions"], ep["rng_seed"], variant=1)
            assert got["completion_time"] == int(ep["completion"]) == int(ep["final"][0])
        else:
            got = H.play_oracle(a, a.x, ep["actions"], ep["rng_seed"], variant=5 if suite in H.SOD_SUITES else 0)
        tag = "%s episode %d (%s)" % (suite, e, a.name)
        assert got["T"] == ep["T"], tag
        for key in ("k", "m", "job_r", "job_n", "done", "step_time", "delay"):
            assert np.array_equal(got[key].astype(np.int64), ep[key].astype(np.int64)), tag + " " + key
        assert np.array_equal(H.bits(got["reward"]), H.bits(ep["reward"])), tag + " reward"
        assert np.array_equal(H.bits(got["state0"]), H.bits(ep["state0"])), tag + " reset state"
        assert np.array_equal(H.states_digest(got["states"]), ep["states_sha256"]), tag + " states digest"
        assert np.array_equal(H.bits(got["states"][-1]), H.bits(ep["state_last"])), tag + " last state"
        if "states" in ep:
            assert np.array_equal(H.bits(got["states"]), H.bits(ep["states"])), tag + " states"
        assert np.array_equal(got["tend"], ep["tend"]), tag + " machine time_end"
        assert got["makespan"] == ep["final"][0] and got["delay_time_sum"] == ep["final"][1], tag


def test_known_reference_values(built):
    """Mk01, fixed rule pair (2, 0): the values the reference produced under the product's x."""
    insts, eps, _ = H.load_suite("mk01")
    ep = eps[2 * 5 + 0]
    assert tuple(ep["actions"][0]) == (2, 0)
    assert ep["T"] == 55                       # one step per operation
    assert -int(ep["reward"].sum()) == int(ep["final"][1])   # rewards telescope to -delay_time_sum
    assert int(ep["final"][0]) == int(ep["tend"].max())


def test_mo_rewards_follow_the_weighting(built):
    """MO_FJSSP_discretes.py:232-244: single-objective weights give integer objective deltas that telescope."""
    _, eps, _ = H.load_suite("mo_discretes")
    seen = set()
    for ep in eps:
        w0, w1, cn, tn = ep["mo"]
        if cn <= 0 and w1 == 1:
            assert -int(ep["reward"].sum()) == int(ep["final"][1]); seen.add("tardiness")
        if cn <= 0 and w0 == 1:
            assert -int(ep["reward"].sum()) == int(ep["completion"]); seen.add("completion")
        if cn > 0:
            want = -(int(ep["completion"]) / cn * w0 + int(ep["final"][1]) / tn * w1)
            assert abs(ep["reward"].sum() - want) < 1e-9 * max(1.0, abs(want)); seen.add("weighted")
    assert seen == {"tardiness", "completion", "weighted"}


def test_sfjsp_reward_is_scaled_makespan(built):
    """SO_SFJSP.py:216-220: rewards sum to -makespan / fluid_completed_time; tardiness counts finished jobs only."""
    insts, eps, _ = H.load_suite("so_sfjsp")
    for ep in eps:
        a = insts[ep["inst"]]
        got = H.play_oracle(a, a.x, ep["actions"], ep["rng_seed"], variant=1)
        want = -int(ep["completion"]) / got["fluid_completed_time"]
        assert abs(ep["reward"].sum() - want) < 1e-9 * max(1.0, abs(want))


def test_reward_telescopes_everywhere(built):
    for suite in H.SUITES:
        _, eps, _ = H.load_suite(suite)
        for ep in eps:
            assert -int(ep["reward"].sum()) == int(ep["final"][1])
            assert bool(ep["done"][-1]) and not ep["done"][:-1].any()
