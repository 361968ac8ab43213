"""GPU: the C-ABI rollout buffer and the batched MPPPO loop on top of the HIP environment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.mark.parametrize("T", [9, 58, 70])     # (episodes of <= 64 steps: the register-resident normalisation kernel; longer: the memory walk)
def test_rollout_buffer_append_and_returns(torch_gpu, T):
    """fjsp_rollout_append == Buffer.py:41-45 `.float()` rows; fjsp_rollout_returns == MPPPO.py:301-312 scan."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.Buffer import RolloutBuffer
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import discounted_returns
    N, S = 37, 20
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
        done = torch.maximum(done, (torch.rand(N, device="cuda", generator=g) < (0.2 if T < 20 else 0.03)).to(torch.uint8))
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
    # scan + per-episode normalisation (MPPPO.py:258-261) in one launch == the tensor-op version on the same returns
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import normalise_returns
    for normalized, standardized in ((True, True), (True, False), (False, True), (False, False)):
        gn = buf.normalised_returns(0.99, normalized, standardized)
        wn = normalise_returns(want, buf.valid[:n].cpu(), normalized, standardized)
        assert torch.equal(buf.returns[:n].cpu(), want)                              # the raw scan is still there
        np.testing.assert_allclose(gn.cpu().numpy(), wn.numpy(), rtol=2e-5, atol=2e-6, err_msg="normalized=%s standardized=%s" % (normalized, standardized))
    buf.clear()
    assert len(buf) == 0
    for _ in range(T):
        buf.add_experience(*rows[0])
    from deep_reinforcement_learning_for_fjsp_amd._capi import FjspError
    with pytest.raises(FjspError):
        buf.add_experience(*rows[0])        # full


def test_policy_sample_kernel(torch_gpu):
    """fjsp_policy_sample == pick_action_and_log_prob (MPPPO.py:272-284): the action frequencies follow the
    probabilities, the log-probability is torch's Categorical.log_prob of the taken action, the pair encoding is
    (a // div, a % div), epsilon = 1 gives uniform actions, and a new seed gives a new stream."""
    torch = torch_gpu
    from torch.distributions import Categorical
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import FusedSampler
    N, A, T = 8192, 30, 4
    torch.manual_seed(0)
    probs = torch.softmax(torch.randn(1, A, device="cuda") * 1.5, -1).repeat(N, 1).contiguous()
    fs = FusedSampler(N, T, A, 5, torch.device("cuda", 0))
    logp = torch.zeros(T, N, device="cuda")
    fs.new_round(0.0)
    counts = torch.zeros(A, device="cuda")
    for t in range(T):
        pair = fs.sample(probs, t, logp[t]).clone()
        a = fs.flat_actions[t].long()
        assert torch.equal(pair[:, 0].long(), a // 5) and torch.equal(pair[:, 1].long(), a % 5)
        want = Categorical(probs, validate_args=False).log_prob(a)
        torch.testing.assert_close(logp[t], want, rtol=1e-5, atol=1e-6)
        counts += torch.bincount(a, minlength=A).float()
    freq = (counts / counts.sum()).cpu().numpy()
    assert np.abs(freq - probs[0].cpu().numpy()).max() < 0.01                 # 32k samples
    first = fs.flat_actions[0].clone()
    assert not torch.equal(first, fs.flat_actions[1])                         # the step counter moves the stream
    fs.new_round(0.0); fs.sample(probs, 0, logp[0])
    assert not torch.equal(first, fs.flat_actions[0])                         # and so does the round seed
    fs.new_round(1.0); fs.sample(probs, 0, logp[0])
    uni = torch.bincount(fs.flat_actions[0].long(), minlength=A).float() / N
    assert float((uni - 1.0 / A).abs().max()) < 0.01
    flat = FusedSampler(N, T, 18, 0, torch.device("cuda", 0))
    flat.new_round(0.0)
    pr = torch.softmax(torch.randn(N, 18, device="cuda"), -1)
    pair = flat.sample(pr, 0, logp[0])
    assert torch.equal(pair[:, 0].long(), flat.flat_actions[0].long()) and int(pair[:, 1].sum()) == 0


@pytest.mark.parametrize("use_graph,fused", [(False, False), (True, False), (False, True), (True, True)])
def test_batched_ppo_rounds_run_on_the_hip_environment(torch_gpu, use_graph, fused):
    """BASELINE config 3 in miniature: 256 envs, actor/critic 2x128, three learning rounds; eager rollout and the
    rollout replayed from a captured HIP graph (policy + env kernel + buffer append, MPPPO.GraphedRollout)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import PPO
    N = 256
    s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    env = BatchedSOFJSSP(s, rng_seed=3)
    torch.manual_seed(0)
    agent = PPO(env, hidden_size=128, hidden_layer=2, seed=1, max_steps=56, use_graph=use_graph, fused_sampling=fused)
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


def test_config3_full_size_ppo_round_on_4096_envs(torch_gpu):
    """BASELINE configs[2] at its full size: 4096 parallel 10x5 SO_FJSSP envs, actor/critic 2x128, the rollout
    replayed from a captured HIP graph with the fused sampler (the shipped configuration of examples/train_ppo.py),
    two learning rounds.  Same invariants as the miniature above, on every one of the 4096 environments."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import PPO
    N = 4096
    s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    env = BatchedSOFJSSP(s, rng_seed=3)
    torch.manual_seed(0)
    agent = PPO(env, hidden_size=128, hidden_layer=2, seed=1, max_steps=56, use_graph=True, fused_sampling=True)
    K = np.array([s.dims(i)["K"] for i in range(N)])
    before = [p.detach().clone() for p in agent.learner.actor_new.parameters()] if hasattr(agent, "learner") else None
    for rnd in range(2):
        tard, mk, (c_loss, a_loss) = agent.run_one_policy_network()
        assert np.isfinite(tard) and np.isfinite(mk) and np.isfinite(c_loss) and np.isfinite(a_loss)
        r = env.read()
        # (only the "stepped after done" bit may be set: the vector loop runs max_steps launches for every env)
        assert bool((r["done"] == 1).all()) and int(((r["status"] & ~4) != 0).sum()) == 0
        assert np.array_equal(r["step_count"].cpu().numpy(), K)
        n = len(agent.memory)
        valid = agent.memory.valid[:n]
        assert np.array_equal(valid.sum(0).cpu().numpy().astype(np.int64), K)    # one valid row per operation
        tot = (agent.memory.rewards[:n].double() * valid.double()).sum(0)
        assert torch.equal(-tot.long(), r["delay_time_sum"])                     # rewards telescope to -tardiness
    assert agent.global_step_number == 2 * int(K.sum())
    if before is not None:
        assert any(not torch.equal(b, p) for b, p in zip(before, agent.learner.actor_new.parameters()))


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


def _dyn_instances(n, seed0, S=2, M=6, max_windows=2):
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    s = fi.InstanceSet(n)
    prm = fi.GenParams(R_min=3, R_max=4, J_min=2, J_max=3, M=M, p_min=5, p_max=40, N_min=1, N_max=3, S=S, DDT=1.0,
                       t_si_min=100.0, t_si_max=200.0)
    for i in range(n):
        seed = seed0 + i
        while True:
            s.generate(i, seed, prm)
            p = np.asarray(s.arrays(i).p)
            if (p.reshape(-1, M) > 0).any(axis=0).all():     # the reference divides by a machine's operation count
                break
            seed += 1000003
        s.generate_machine_data(i, seed, max_windows=max_windows, window_gap=(10, 80), window_len=(3, 20))
    return s.solve_fluid()


def test_ddqn_rounds_on_so_sfjsp(torch_gpu):
    """agents/DDQN/DDQN.py end to end on batches of the environment it instantiates (SO_SFJSP): replay ring
    fills with one transition per operation, the double-Q update moves the local net, the target follows by tau."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOSFJSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.DDQN.DDQN import DDQN
    N = 128
    test_set = fi.InstanceSet(8).generate_range(7000, fi.bench_10x5_params()).solve_fluid()
    test_env = BatchedSOSFJSP(test_set, rng_seed=5)
    rounds = [0]

    def make_train_env():
        rounds[0] += 1
        s = fi.InstanceSet(N).generate_range(20000 * rounds[0], fi.bench_10x5_params()).solve_fluid()
        return BatchedSOSFJSP(s, rng_seed=rounds[0])

    torch.manual_seed(0)
    agent = DDQN(make_train_env, test_env, hidden_size=64, hidden_layer=2, seed=3, updates_per_round=4,
                 hyper={"batch_size": 512, "buffer_size": 20000, "learning_rate": 1e-3, "num_episodes_to_run": 20})
    before = [p.detach().clone() for p in agent.q_network_local.parameters()]
    tgt_before = [p.detach().clone() for p in agent.q_network_target.parameters()]
    tests = [agent.step() for _ in range(3)]
    assert all(np.isfinite(t) and t > 0 for t in tests)
    n_ops = agent.global_step_number
    assert len(agent.memory) == min(n_ops, 20000) and n_ops > 3 * N * 25
    assert agent.last_loss is not None and np.isfinite(agent.last_loss)
    assert any(not torch.equal(b, p) for b, p in zip(before, agent.q_network_local.parameters()))
    moved = [float((p.detach() - b).abs().max()) for b, p in zip(tgt_before, agent.q_network_target.parameters())]
    assert 0 < max(moved) < 0.1                                            # soft update, tau = 0.005
    assert agent.completed_time == min(tests) and agent.best_state_dict is not None
    assert 0.01 <= agent.exploration_strategy.epsilon < 1.0                # decayed once per vector pick


def test_da3c_and_sac_controller_on_mo_dfjsp(torch_gpu, tmp_path):
    """HMPSAC end to end on the dynamic environment: three lower-level objective policies trained by the
    batched double-actor A2C (A3C_v5.{1,2,3}.py), saved/loaded through the reference's checkpoint layout, then
    the SAC-discrete controller choosing among them (SAC_Discrete.py:197-240)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMODFJSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import DA3C
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.SAC_Discrete import SAC_Discrete
    N = 48
    test_env = BatchedMODFJSP(_dyn_instances(6, 400), rng_seed=2)
    rounds = [0]

    def make_train_env():
        rounds[0] += 1
        return BatchedMODFJSP(_dyn_instances(N, 1000 * rounds[0]), rng_seed=rounds[0])

    torch.manual_seed(0)
    for policy in (0, 1, 2):
        tr = DA3C(make_train_env, test_env, reward_policy=policy, hidden_size=32, hidden_layer=2, seed=policy, max_steps=400)
        before = [p.detach().clone() for p in tr.actor_machine_model.parameters()]
        objs = [tr.run_one_round() for _ in range(2)]
        assert all(np.isfinite(o) and o >= 0 for o in objs) and all(np.isfinite(v) for v in tr.last_losses)
        assert any(not torch.equal(b, p) for b, p in zip(before, tr.actor_machine_model.parameters()))
        assert tr.objective_min == min(objs)
        folder = tmp_path / ("policy_networks_v5.%d" % (policy + 1))
        folder.mkdir()
        tr.save_actor_model(str(folder))
    # the controller loads reference-shaped nets (3 x 200) from that layout: retrain one quickly at that size
    big = DA3C(make_train_env, test_env, reward_policy=0, seed=9, max_steps=400)
    for policy in (0, 1, 2):
        big.save_actor_model(str(tmp_path / ("policy_networks_v5.%d" % (policy + 1))))
    env = BatchedMODFJSP(_dyn_instances(N, 77), rng_seed=4)
    sac = SAC_Discrete(env, lower_policies=str(tmp_path), hidden_size=32, hidden_layer=2, seed=1, max_steps=400,
                       hyper={"min_steps_before_learning": 300, "update_every_n_steps": 200, "batch_size": 128,
                              "learning_updates_per_learning_session": 2})
    actor_before = [p.detach().clone() for p in sac.actor_local.parameters()]
    out = sac.run_n_episodes(2)
    assert len(out) == 2 and all(np.isfinite(v) and v >= 0 for ep in out for v in ep)
    assert sac.global_step_number == len(sac.memory) or len(sac.memory) == sac.memory.capacity
    assert sac.learn_sessions > 0 and all(np.isfinite(v) for v in sac.last_losses)
    assert any(not torch.equal(b, p) for b, p in zip(actor_before, sac.actor_local.parameters()))
    assert float(sac.alpha) > 0 and float(sac.alpha) != 1.0                     # entropy temperature was tuned
    st = env.read()["status"].cpu().numpy()
    assert ((st & ~4) == 0).all()
    rw = sac.memory.rewards[:len(sac.memory)]
    assert bool(torch.isfinite(rw).all()) and float(rw.max()) <= 0.0          # every objective only grows


def test_config5_full_size_sac_controller_on_4096_dynamic_envs(torch_gpu, tmp_path):
    """BASELINE configs[4] at its full size on the environment side: 4096 MO_DFJSP_breakdown envs (two orders:
    every episode re-solves the fluid LP at the second order's arrival, breakdown windows on every machine), the
    SAC-discrete controller choosing among three lower policies of the reference's shape for one full episode
    of every env.  Every env finishes, every arrival got its LP, rewards only ever decrease the objectives."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMODFJSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import DA3C
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.SAC_Discrete import SAC_Discrete
    N = 4096
    small = BatchedMODFJSP(_dyn_instances(8, 400), rng_seed=2)
    torch.manual_seed(0)
    big = DA3C(lambda: small, small, reward_policy=0, seed=9, max_steps=400)      # reference-shaped (3 x 200) lower nets
    for policy in (0, 1, 2):
        folder = tmp_path / ("policy_networks_v5.%d" % (policy + 1))
        folder.mkdir()
        big.save_actor_model(str(folder))
    insts = _dyn_instances(N, 9000)
    env = BatchedMODFJSP(insts, rng_seed=4)
    sac = SAC_Discrete(env, lower_policies=str(tmp_path), hidden_size=32, hidden_layer=2, seed=1, max_steps=400,
                       hyper={"min_steps_before_learning": 20000, "update_every_n_steps": 40000, "batch_size": 256,
                              "learning_updates_per_learning_session": 1})
    out = sac.run_n_episodes(1)
    assert len(out) == 1 and all(np.isfinite(v) and v >= 0 for v in out[0])
    r = env.read()
    assert bool((r["done"] == 1).all())
    assert ((r["status"].cpu().numpy() & ~4) == 0).all()
    n_ops = np.array([int((insts.arrays(i).count.sum(0) * insts.arrays(i).Jr).sum()) for i in range(0, N, 64)])
    assert np.array_equal(r["step_count"].cpu().numpy()[::64], n_ops)             # one step per operation of every order
    # one LP per env and episode played (the second order's arrival; reset-time LPs are solved with the instances);
    # the controller plays the three lower policies once for its normalisers before its own episode
    assert env.batch.lp_solves >= N and env.batch.lp_solves % N == 0
    assert sac.learn_sessions > 0 and all(np.isfinite(v) for v in sac.last_losses)
    rw = sac.memory.rewards[:len(sac.memory)]
    assert bool(torch.isfinite(rw).all()) and float(rw.max()) <= 0.0
    assert bool((r["energy_consumption"] > 0).all()) and bool((r["makespan"] > 0).all())


def test_da3c_on_so_dfjsp(torch_gpu):
    """agents/DA3C/DA3C_double_actor.py: the same double-actor A2C at sizes 20 -> 6 / 21 -> 5 on its environment,
    environments/SO_DFJSP.py (order arrivals, due date = order delivery)."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSODFJSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import DA3C
    prm = fi.GenParams(R_min=3, R_max=4, J_min=2, J_max=3, M=5, p_min=5, p_max=40, N_min=1, N_max=3, S=2, DDT=1.0,
                       t_si_min=100.0, t_si_max=200.0)
    test_env = BatchedSODFJSP(fi.InstanceSet(6).generate_range(300, prm).solve_fluid(), rng_seed=2)
    rounds = [0]

    def make_train_env():
        rounds[0] += 1
        return BatchedSODFJSP(fi.InstanceSet(32).generate_range(5000 * rounds[0], prm).solve_fluid(), rng_seed=rounds[0])

    torch.manual_seed(0)
    tr = DA3C(make_train_env, test_env, reward_policy=None, hidden_size=32, hidden_layer=2, seed=1, max_steps=300,
              state_size=20, actions_size=(6, 5))
    before = [p.detach().clone() for p in tr.actor_task_model.parameters()]
    objs = [tr.run_one_round() for _ in range(2)]
    assert all(np.isfinite(o) and o >= 0 for o in objs) and all(np.isfinite(v) for v in tr.last_losses)
    assert any(not torch.equal(b, p) for b, p in zip(before, tr.actor_task_model.parameters()))
    assert tr.actor_machine_model.layers_2[0].in_features == 21 and tr.actor_task_model.layers_1[-1].out_features == 6



def test_native_actor_matches_the_torch_actor(torch_gpu):
    """fjsp_actor_forward (the actor MLP as the fused policy rollout evaluates it: f32 fmaf chains in a fixed order,
    weights in LDS) against ActorNet.forward in PyTorch-ROCm f32 on the same weights and states: 1e-5 relative on the
    probabilities (tolerance of a differently ordered f32 sum over <= 128 terms), rows sum to 1."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import ActorNet, native_actor_forward, native_actor_params
    torch.manual_seed(3)
    for S, A, n in ((20, 30, 1000), (25, 18, 37), (18, 20, 16)):
        net = ActorNet(S, 128, 2, A).cuda()
        states = (torch.randn(n, S, dtype=torch.float64, device="cuda") * 3.0)
        with torch.no_grad():
            want = net(states.float())
        got = native_actor_forward(net, states)
        assert got.shape == want.shape
        assert float((got.sum(1) - 1).abs().max()) < 1e-5
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert native_actor_params(ActorNet(20, 64, 2, 30).cuda()) is None          # other shapes: no in-kernel path
    assert native_actor_params(ActorNet(20, 128, 3, 30).cuda()) is None


@pytest.mark.parametrize("which", ["so_fjssp", "mo_discretes"])
def test_fused_policy_rollout_equals_the_per_step_path(torch_gpu, which):
    """fjsp_env_rollout_policy (actor inside the environment kernel, one launch per rollout) against the per-step
    loop it replaces (MPPPO.py:245-252 as actor kernel -> fjsp_policy_sample -> fjsp_env_step -> fjsp_rollout_append)
    with the same weights, seed and exploration rate: sampled actions, log-probabilities, rewards, states, next
    states, done flags and the valid mask of every row, and the environments' final attributes -- bit for bit."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMOFJSSP, BatchedSOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.Buffer import RolloutBuffer
    N, T = 150, 56                                    # (not a multiple of 16: the last workgroup is partial)
    insts = fi.InstanceSet(N).generate_range(3000, fi.bench_10x5_params()).solve_fluid()
    if which == "so_fjssp":
        mk = lambda: BatchedSOFJSSP(insts, rng_seed=9)
        S, A, div = 20, 30, 5
    else:
        def mk():
            e = BatchedMOFJSSP(insts, rng_seed=9)
            e.set_objective((0.5, 0.5), torch.full((N,), 60.0, dtype=torch.float64), torch.full((N,), 400.0, dtype=torch.float64))
            return e
        S, A, div = 25, 18, 0
    torch.manual_seed(11)
    learner = M.PPOLearner(S, A, 128, 2, 2, device=torch.device("cuda", 0), seed=5)
    out = {}
    for mode in ("per_step", "fused"):
        env = mk()
        memory = RolloutBuffer(T, N, S, device=0)
        fused = M.FusedSampler(N, T, A, div, torch.device("cuda", 0))
        fused.rounds = 41                              # the same sampling stream for both runs
        old_log_prob = torch.zeros(T, N, device="cuda")
        torch.manual_seed(77)
        if mode == "per_step":
            fused.native_actor = True
            fused.new_round(0.15)
            M._rollout_body(env, learner, memory, old_log_prob, 0.15, T, lambda a: a, None, False, fused)
        else:
            assert M.fused_policy_rollout(env, learner, memory, fused, old_log_prob, 0.15, T)
        assert len(memory) == T
        torch.cuda.synchronize()
        valid = memory.valid[:T].clone()
        r = env.read()
        keep = lambda x: torch.where((valid if x.dim() == 2 else valid.unsqueeze(-1)) > 0, x, torch.zeros_like(x))   # valid rows only
        assert all(bool(torch.isfinite(x).all()) for x in (memory.states[:T], memory.next_states[:T], old_log_prob))
        out[mode] = dict(valid=valid, flat=keep(fused.flat_actions[:T]), logp=keep(old_log_prob),
                         states=keep(memory.states[:T]), nxt=keep(memory.next_states[:T]),
                         rewards=keep(memory.rewards[:T]), dones=keep(memory.dones[:T]), actions=keep(memory.actions[:T]),
                         final_state=env.batch.state.clone(), delay=r["delay_time_sum"], makespan=r["makespan"], steps=r["step_count"],
                         status=r["status"] & ~4)
    a, b = out["per_step"], out["fused"]
    assert int(a["valid"].sum()) == int(a["steps"].sum()) and bool((a["steps"] > 0).all())
    for k in a:
        if not torch.equal(a[k], b[k]):
            bad = (a[k] != b[k]).nonzero()
            raise AssertionError("%s differs at %d places, first %s: %r vs %r" % (
                k, bad.shape[0], bad[0].tolist(), a[k][tuple(bad[0].tolist())].item(), b[k][tuple(bad[0].tolist())].item()))
    assert int(a["status"].abs().sum()) == 0
    assert len(set(a["flat"].long().flatten().tolist())) > A // 2              # the policy did sample around


def test_ppo_rounds_with_the_fused_policy_rollout(torch_gpu):
    """PPO(fused_rollout=True): learning rounds on top of the one-launch rollout, same invariants as the per-step agent."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import PPO
    N = 1024
    s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
    env = BatchedSOFJSSP(s, rng_seed=3)
    torch.manual_seed(0)
    agent = PPO(env, hidden_size=128, hidden_layer=2, seed=1, max_steps=56, use_graph=True, fused_rollout=True)
    K = np.array([s.dims(i)["K"] for i in range(N)])
    before = [p.detach().clone() for p in agent.learner.actor_new.parameters()]
    for rnd in range(3):
        tard, mk, (c_loss, a_loss) = agent.run_one_policy_network()
        assert np.isfinite(tard) and np.isfinite(mk) and np.isfinite(c_loss) and np.isfinite(a_loss)
        r = env.read()
        assert bool((r["done"] == 1).all()) and int(((r["status"] & ~4) != 0).sum()) == 0
        assert np.array_equal(r["step_count"].cpu().numpy(), K)
        valid = agent.memory.valid[:len(agent.memory)]
        assert np.array_equal(valid.sum(0).cpu().numpy().astype(np.int64), K)
        tot = (agent.memory.rewards[:len(agent.memory)].double() * valid.double()).sum(0)
        assert torch.equal(-tot.long(), r["delay_time_sum"])
    assert any(not torch.equal(b, p) for b, p in zip(before, agent.learner.actor_new.parameters()))


def test_fused_ppo_learning_matches_autograd(torch_gpu):
    """agents/fused_mlp.py (library GEMMs + csrc/fjsp_ppo.hip: loss and its gradient, ReLU backward + bias gradient,
    clip + Adam on flat buffers) against the autograd path of PPOLearner on the same 40 000 samples and weights:
    one learning iteration (tight: f32 reassociation only) and a full 10-iteration round."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import PPOLearner
    n, S, A = 40000, 20, 30
    g = torch.Generator(device="cuda").manual_seed(5)
    states = torch.randn(n, S, device="cuda", generator=g)
    actions = torch.randint(0, A, (n,), device="cuda", generator=g)
    old_lp = -torch.rand(n, device="cuda", generator=g) * 4 - 0.5
    returns = torch.randn(n, device="cuda", generator=g)
    valid = torch.ones(n, device="cuda")
    for iters, rtol, atol in ((1, 2e-4, 2e-6), (10, 2e-2, 2e-4)):
        hyper = {"learning_iterations_per_round_critic": iters, "learning_iterations_per_round_actor": iters}
        ref = PPOLearner(S, A, 128, 2, 2, device="cuda", seed=9, hyper=hyper); ref.fused_learn = False
        fus = PPOLearner(S, A, 128, 2, 2, device="cuda", seed=9, hyper=hyper)
        for pr, pf in zip(list(ref.actor_new.parameters()) + list(ref.critic.parameters()),
                          list(fus.actor_new.parameters()) + list(fus.critic.parameters())):
            assert torch.equal(pr, pf)
        lr_ = ref.learn(states, actions, old_lp, returns, valid)
        lf_ = fus.learn(states, actions, old_lp, returns, valid)
        assert fus._use_fused and not ref._use_fused
        np.testing.assert_allclose(lf_, lr_, rtol=max(rtol, 1e-4), atol=1e-5)
        for name, nr, nf in (("actor", ref.actor_new, fus.actor_new), ("critic", ref.critic, fus.critic), ("actor_old", ref.actor_old, fus.actor_old)):
            for pr, pf in zip(nr.parameters(), nf.parameters()):
                np.testing.assert_allclose(pf.detach().cpu().numpy(), pr.detach().cpu().numpy(), rtol=rtol, atol=atol,
                                           err_msg="%s after %d iteration(s)" % (name, iters))
    # the trainer path is fixed at the first learn(): a later, smaller batch stays on the fused trainer (one set of Adam
    # moments), it does not wake the torch optimisers up
    small = slice(0, 2000)
    fus.learn(states[small], actions[small], old_lp[small], returns[small], valid[small])
    assert fus._path == "fused" and fus._use_fused and len(fus.actor_optimizer.state) == 0 and len(fus.critic_optimizer.state) == 0
    ref.learn(states, actions, old_lp, returns, valid)
    assert ref._path == "eager" and not ref._use_fused
    # the parameters live in ONE flat buffer (the all-reduce bucket), the module still owns them
    actor_tr, _ = fus._fused_nets()
    assert actor_tr.flat.numel() == sum(p.numel() for p in fus.actor_new.parameters())
    assert fus.actor_new.layers[0].weight.data_ptr() == actor_tr.flat.data_ptr()


@pytest.mark.gpu
@pytest.mark.parametrize("n,S,A", [(1000, 20, 24), (32 * 300 + 7, 18, 30), (5, 30, 32), (70000, 25, 6)])
def test_one_launch_training_pass_matches_autograd(torch_gpu, n, S, A):
    """fjsp_mlp_train_pass (csrc/fjsp_mlp_train.hip: forward + loss + backward of a 2 x 128 network in one launch on the f32
    matrix cores) against autograd on the same weights and samples: loss and every parameter gradient, actor and critic,
    sample counts that are not a multiple of the 32-sample tile.  Tolerance: f32 reassociation of sums over n samples."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
    from deep_reinforcement_learning_for_fjsp_amd.agents import fused_mlp
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(n + S)
    torch.manual_seed(3)
    actor, critic = M.ActorNet(S, 128, 2, A).to(dev), M.CriticNet(S, 128, 2, 1).to(dev)
    with torch.no_grad():
        for net in (actor, critic):
            for prm in net.parameters():
                if prm.dim() == 1:
                    prm.copy_(torch.randn(prm.shape, device=dev, generator=g) * 0.1)       # biases away from 0: the bias path is exercised
    x = torch.randn(n, S, device=dev, generator=g)
    actions = torch.randint(0, A, (n,), device=dev, generator=g)
    old_lp = -torch.rand(n, device=dev, generator=g) * 3 - 0.2
    adv = torch.randn(n, device=dev, generator=g)
    ret = torch.randn(n, device=dev, generator=g)
    count = torch.full((1,), float(n), device=dev)
    # autograd
    new_lp = actor.log_prob(x, actions)
    a_loss = -(M.actor_loss_terms(new_lp, old_lp, adv, 0.2)).sum() / n
    a_grads = torch.autograd.grad(a_loss, list(actor.parameters()))
    c_loss = ((critic(x).squeeze(1) - ret) ** 2).sum() / n
    c_grads = torch.autograd.grad(c_loss, list(critic.parameters()))
    for net, mode, loss_ref, grads_ref, aux in ((actor, 0, a_loss, a_grads, (actions.float(), old_lp, adv)), (critic, 1, c_loss, c_grads, (ret, None, None))):
        tr = fused_mlp.FusedMLP(net.layers, lr=1e-3)
        assert tr.mfma_pass_supported()
        loss = tr.train_pass(mode, x, aux[0], aux[1], aux[2], count, 0.2)
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(loss), float(loss_ref), rtol=2e-5, atol=1e-7)
        for view, ref in zip(tr.views, grads_ref):
            scale = float(ref.abs().max()) + 1e-12
            np.testing.assert_allclose(view.cpu().numpy(), ref.cpu().numpy(), rtol=1e-3, atol=2e-5 * scale,
                                       err_msg="mode %d, gradient of shape %s" % (mode, tuple(ref.shape)))
        # deterministic: a second launch gives the same bits
        first = tr.grad.clone()
        tr.train_pass(mode, x, aux[0], aux[1], aux[2], count, 0.2)
        assert torch.equal(first, tr.grad)


@pytest.mark.gpu
def test_train_step_equals_pass_plus_optimiser_step(torch_gpu):
    """fjsp_mlp_train_step (pass, gradient finish + squared norm + step count, clip + Adam: three launches) against
    train_pass() followed by the separate fjsp_adam_clip_step, two steps on the same weights and samples: identical
    gradients, parameters equal up to the summation order of the clip norm."""
    torch = torch_gpu
    import copy
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
    from deep_reinforcement_learning_for_fjsp_amd.agents import fused_mlp
    dev = torch.device("cuda", 0)
    n, S, A = 5000, 20, 24
    g = torch.Generator(device="cuda").manual_seed(11)
    torch.manual_seed(5)
    net_a = M.ActorNet(S, 128, 2, A).to(dev)
    net_b = copy.deepcopy(net_a)
    x = torch.randn(n, S, device=dev, generator=g)
    actions = torch.randint(0, A, (n,), device=dev, generator=g).float()
    old_lp = -torch.rand(n, device=dev, generator=g) * 3 - 0.2
    adv = torch.randn(n, device=dev, generator=g) * 50.0            # large advantages: the clip coefficient is < 1
    count = torch.full((1,), float(n), device=dev)
    ta, tb = fused_mlp.FusedMLP(net_a.layers, lr=1e-3, max_norm=1.0), fused_mlp.FusedMLP(net_b.layers, lr=1e-3, max_norm=1.0)
    for _ in range(2):
        la = ta.train_pass(0, x, actions, old_lp, adv, count, 0.2).clone()
        ga = ta.grad.clone()
        ta.step()
        lb = tb.train_step(0, x, actions, old_lp, adv, count, 0.2).clone()
        torch.cuda.synchronize()
        np.testing.assert_allclose(tb.grad.cpu().numpy(), ga.cpu().numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(float(lb), float(la), rtol=1e-6)
        assert float(tb.step_count) == float(ta.step_count)
        np.testing.assert_allclose(tb.flat.cpu().numpy(), ta.flat.cpu().numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,hidden,layers", [(4096, 200, 3), (37, 200, 4), (1000, 64, 1), (9, 256, 2)])
def test_one_launch_policy_pair_sampling(torch_gpu, rows, hidden, layers):
    """fjsp_policy_pair_sample == SAC_Discrete.py:277-284 (TaskPolicyNet -> Categorical -> MachinePolicyNet on
    cat(state, a_t) -> Categorical): the probabilities the kernel draws from equal the torch networks' softmax (f32
    reassociation: 2e-5), every action is the inverse-CDF draw of the documented splitmix64 stream from exactly those
    probabilities (host restatement), draw counters advance so a second call draws differently, and the action
    frequencies follow the probabilities."""
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import MachinePolicyNet, TaskPolicyNet
    from deep_reinforcement_learning_for_fjsp_amd.agents import fused_policy
    torch.manual_seed(5)
    S = 30
    task = TaskPolicyNet(S, hidden, layers, 12).cuda()
    machine = MachinePolicyNet(S + 1, hidden, layers, 10).cuda()
    with torch.no_grad():                                   # (sharper distributions than the initialisation's near-uniform ones)
        for net in (task.layers_1, machine.layers_2):
            net[-1].weight.mul_(6.0)
    state = (torch.rand(rows, S, dtype=torch.float64, device="cuda") * 4.0 - 1.0).contiguous()
    assert fused_policy.supported(task.layers_1, "cuda:0") and fused_policy.supported(machine.layers_2, "cuda:0")
    sm = fused_policy.PolicyPairSampler(task.layers_1, machine.layers_2, seed=1234)
    a_t, a_m, p_t, p_m = sm.sample(state, probs=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        want_t = task(state.float())
        want_m = machine(torch.cat([state.float(), a_t.float().unsqueeze(1)], 1))
    assert torch.allclose(p_t, want_t, rtol=2e-5, atol=1e-7)
    assert torch.allclose(p_m, want_m, rtol=2e-5, atol=1e-7)
    pt, pm, at, am = p_t.cpu().numpy(), p_m.cpu().numpy(), a_t.cpu().numpy(), a_m.cpu().numpy()
    for r in list(range(min(rows, 64))) + [rows - 1]:
        assert at[r] == fused_policy.expected_draw(pt[r], 1234, r, 0), r
        assert am[r] == fused_policy.expected_draw(pm[r], 1234, r, 1), r
    assert int(sm.draws(rows).min()) == 2 and int(sm.draws(rows).max()) == 2
    b_t, b_m, q_t, _ = sm.sample(state, probs=True)
    assert torch.equal(q_t, p_t)
    bt = b_t.cpu().numpy()
    for r in range(min(rows, 32)):
        assert bt[r] == fused_policy.expected_draw(pt[r], 1234, r, 2), r
    if rows >= 1000:
        assert not torch.equal(a_t, b_t)
        # frequencies of 200 draws of the first 64 rows against their probabilities
        rep = state[:64].repeat(1, 1).contiguous()
        sm2 = fused_policy.PolicyPairSampler(task.layers_1, None, seed=77)
        counts = torch.zeros(64, 12, device="cuda")
        for _ in range(200):
            t, none = sm2.sample(rep)
            assert none is None
            counts.scatter_add_(1, t.unsqueeze(1), torch.ones(64, 1, device="cuda"))
        freq = counts / 200.0
        assert float((freq - want_t[:64]).abs().max()) < 0.15
    # the action pair in the environment's encoding, written by the kernel itself; with `select` only the chosen rows
    sm3 = fused_policy.PolicyPairSampler(task.layers_1, machine.layers_2, seed=1234)
    pair = torch.full((rows, 2), 255, dtype=torch.uint8, device="cuda")
    c_t, c_m = sm3.sample(state, pair_out=pair)
    assert torch.equal(c_t, a_t) and torch.equal(c_m, a_m)                   # (same seed, same draw counters as the first call)
    assert torch.equal(pair, torch.stack([a_t, a_m], 1).to(torch.uint8))
    select = (torch.arange(rows, device="cuda") % 3).contiguous()
    pair2 = torch.full((rows, 2), 255, dtype=torch.uint8, device="cuda")
    d_t, d_m = sm3.sample(state, pair_out=pair2, select=select, which=1)
    mine = select == 1
    assert torch.equal(pair2[mine], torch.stack([d_t, d_m], 1).to(torch.uint8)[mine])
    assert bool((pair2[~mine] == 255).all())
    # a network that does not fit is refused (the caller keeps the library path)
    wide = TaskPolicyNet(S, 300, 2, 12).cuda()
    assert not fused_policy.supported(wide.layers_1)
    with pytest.raises(ValueError):
        fused_policy.PolicyPairSampler(wide.layers_1)


@pytest.mark.gpu
def test_critic_step_hands_out_the_values_of_its_forward_pass(torch_gpu):
    """fjsp_mlp_train_step_values: the critic's training step also returns V(s) under the parameters BEFORE the update
    (what MPPPO.py:263 builds the advantages from): equal to the network's own forward pass (f32 reassociation), and the
    step itself is the one fjsp_mlp_train_step takes."""
    import copy
    torch = torch_gpu
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
    from deep_reinforcement_learning_for_fjsp_amd.agents import fused_mlp
    dev = torch.device("cuda", 0)
    n, S = 70001, 20
    g = torch.Generator(device="cuda").manual_seed(3)
    torch.manual_seed(9)
    net_a = M.CriticNet(S, 128, 2, 1).to(dev)
    net_b = copy.deepcopy(net_a)
    x = torch.randn(n, S, device=dev, generator=g)
    returns = torch.randn(n, device=dev, generator=g)
    count = torch.full((1,), float(n), device=dev)
    with torch.no_grad():
        want = net_a(x).squeeze(1).clone()
    ta, tb = fused_mlp.FusedMLP(net_a.layers, lr=1e-3, max_norm=1.0), fused_mlp.FusedMLP(net_b.layers, lr=1e-3, max_norm=1.0)
    values = torch.full((n,), float("nan"), device=dev)
    la = ta.train_step(1, x, returns, None, None, count, values_out=values).clone()
    lb = tb.train_step(1, x, returns, None, None, count).clone()
    torch.cuda.synchronize()
    np.testing.assert_allclose(values.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert float(la) == float(lb)
    assert torch.equal(ta.flat, tb.flat)
    with pytest.raises(ValueError):
        ta.train_step(0, x, returns, returns, returns, count, values_out=values)
