"""GPU: the C-ABI rollout buffer and the batched MPPPO loop on top of the HIP environment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def test_rollout_buffer_append_and_returns(torch_gpu):
    """fjsp_rollout_append == Buffer.py:41-45 `.float()` rows; fjsp_rollout_returns == MPPPO.py:301-312 scan."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.Buffer import RolloutBuffer
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import discounted_returns
    T, N, S = 9, 37, 20
    buf = RolloutBuffer(T, N, S)
    g = torch.Generator(device="cuda").manual_seed(0)
    rows = []
    done = torch.zeros(N, dtype=torch.uint8, device="cuda")
    for t in range(T - 2):
        st = torch.randn(N, S, dtype=torch.float64, device="cuda", generator=g)
        nx = torch.randn(N, S, dtype=torch.float64, device="cuda", generator=g)
        ac = torch.randint(0, 5, (N, 2), dtype=torch.uint8, device="cuda", generator=g)
        rw = -torch.randint(0, 300, (N,), device="cuda", generator=g).double()
        active = (done == 0).to(torch.uint8)
        done = torch.maximum(done, (torch.rand(N, device="cuda", generator=g) < 0.2).to(torch.uint8))
        buf.add_experience(st, ac, rw, nx, done, active)
        rows.append((st, ac, rw, nx, done.clone(), active))
    n = len(buf)
    assert n == T - 2
    states, actions, rewards, nexts, dones = buf.sample()
    for t, (st, ac, rw, nx, dn, active) in enumerate(rows):
        assert torch.equal(states[t], st.float()) and torch.equal(nexts[t], nx.float())
        assert torch.equal(actions[t], ac.float()) and torch.equal(rewards[t], rw.float())
        assert torch.equal(dones[t], dn.float()) and torch.equal(buf.valid[t], active.float())
    got = buf.compute_returns(0.99)
    want = discounted_returns(rewards.cpu(), buf.valid[:n].cpu(), 0.99)
    assert torch.equal(got.cpu(), want)
    buf.clear()
    assert len(buf) == 0
    for _ in range(T):
        buf.add_experience(*rows[0])
    from deep_reinforcement_learning_for_fjsp_amd._capi import FjspError
    with pytest.raises(FjspError):
        buf.add_experience(*rows[0])        # full


def test_batched_ppo_rounds_run_on_the_hip_environment(torch_gpu):
    """BASELINE config 3 in miniature: 256 envs, actor/critic 2x128, three learning rounds."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import PPO
    N = 256
    s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    env = BatchedSOFJSSP(s, rng_seed=3)
    torch.manual_seed(0)
    agent = PPO(env, hidden_size=128, hidden_layer=2, seed=1, max_steps=56)
    K = np.array([s.dims(i)["K"] for i in range(N)])
    for rnd in range(3):
        tard, mk, (c_loss, a_loss) = agent.run_one_policy_network()
        assert np.isfinite(tard) and np.isfinite(mk) and np.isfinite(c_loss) and np.isfinite(a_loss)
        r = env.read()
        assert bool((r["done"] == 1).all())
        assert np.array_equal(r["step_count"].cpu().numpy(), K)
        n = len(agent.memory)
        valid = agent.memory.valid[:n]
        assert np.array_equal(valid.sum(0).cpu().numpy().astype(np.int64), K)    # one valid row per operation
        # rewards of the valid rows telescope to -delay_time_sum
        tot = (agent.memory.rewards[:n].double() * valid.double()).sum(0)
        assert torch.equal(-tot.long(), r["delay_time_sum"])
    assert agent.episode_number == 3 and agent.global_step_number > 0


def test_multi_policy_ppo_on_mo_discretes(torch_gpu):
    """agents/MPPPO/MPPPO.py end to end on the environment it instantiates (MO_FJSSP_discretes):
    5 policies, completion / tardiness normalisers from the single-objective runs, evolution step."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import MPPPO
    N = 128
    test_set = fi.InstanceSet(16).generate_range(5000, fi.bench_10x5_params()).solve_fluid()
    test_env = BatchedMOFJSSP(test_set, rng_seed=1)
    epoch = [0]

    def make_train_env():       # a fresh random batch per epoch, like generated_new_environment() (MPPPO.py:149-154)
        epoch[0] += 1
        s = fi.InstanceSet(N).generate_range(10000 * epoch[0], fi.bench_10x5_params()).solve_fluid()
        return BatchedMOFJSSP(s, rng_seed=epoch[0])

    torch.manual_seed(0)
    agent = MPPPO(make_train_env, test_env, actor_number=5, hidden_size=64, hidden_layer=2, critic_layer=2,
                  max_steps=56, evolve_every=1)
    assert agent.weight_vector_dict[0] == (1.0, 0.0) and agent.weight_vector_dict[4] == (0.0, 1.0)
    before = [p.detach().clone() for p in agent.learners[2].actor_new.parameters()]
    hist = agent.run_n_episodes(2)
    assert len(hist) == 2 and set(hist[0]) == set(range(5))
    for objs in hist:
        for c, t in objs.values():
            assert np.isfinite(c) and np.isfinite(t) and c > 0
    assert np.isfinite(agent.completion_min) and np.isfinite(agent.tardiness_min)
    assert any(not torch.equal(b, p) for b, p in zip(before, agent.learners[2].actor_new.parameters()))
    chosen = agent.multi_policy_update(hist[-1])
    assert set(chosen) == set(range(5)) and all(0 <= v < 5 for v in chosen.values())
