"""The DDQN / SAC-discrete / A3C / PPO counterparts against tests/golden/agent_updates.npz and ppo_update.npz: losses and
every parameter after ONE update (PPO: one learning round), computed by THE REFERENCE'S OWN agent classes at fixed weights
(tests/golden/make_agent_fixtures_ref.py imports agents/DDQN/DDQN.py, agents/HMPSAC/SAC_Discrete.py, A3C_v5.1.py and
agents/MPPPO/MPPPO.py from /root/reference in the build container and calls their learn / loss / optimisation-step methods;
make_agent_fixtures.py, the round-2 transcription, stays as its cross-check).  The same check runs on the CPU (here) and on
the MI355X (-m gpu): same weights, same batch, f32 tolerance."""
import os
import re

import numpy as np
import pytest
import torch

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "agent_updates.npz")
RTOL, ATOL = 2e-5, 2e-6


class _Env(object):
    def __init__(self, device):
        self.device = torch.device(device)
        self.N = 4


def _load(module, fx, prefix, device):
    sd = {}
    for k in module.state_dict():
        fk = "%s/%s" % (prefix, re.sub(r"^layers(_\d)?\.", "layers.", k))
        sd[k] = torch.from_numpy(fx[fk]).to(device)
    module.load_state_dict(sd)


def _assert_module(module, fx, prefix, what):
    for k, v in module.state_dict().items():
        fk = "%s/%s" % (prefix, re.sub(r"^layers(_\d)?\.", "layers.", k))
        want = fx[fk]
        got = v.detach().cpu().numpy()
        if want.dtype.kind in "iu":                    # BatchNorm's num_batches_tracked
            assert np.array_equal(got, want), (what, k)
        else:
            np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL, err_msg="%s %s" % (what, k))


def _t(fx, key, device, dtype=torch.float32):
    return torch.from_numpy(fx[key]).to(device=device, dtype=dtype)


def _ddqn(device):
    from deep_reinforcement_learning_for_fjsp_amd.agents.DDQN.DDQN import DDQN
    fx = np.load(FIX)
    lr, gamma, clip, tau = fx["ddqn/hyper"]
    agent = DDQN(None, _Env(device), hidden_size=16, hidden_layer=2, seed=1,
                 hyper={"learning_rate": float(lr), "discount_rate": float(gamma), "gradient_clipping_norm": float(clip),
                        "tau": float(tau), "batch_size": 32, "buffer_size": 64})
    _load(agent.q_network_local, fx, "ddqn/local0", device)
    _load(agent.q_network_target, fx, "ddqn/target0", device)
    batch = tuple(_t(fx, "ddqn/" + k, device) for k in ("states", "actions", "rewards", "next_states", "dones"))
    loss = agent.learn(experiences=batch)                                                  # DDQN.py:168-180
    np.testing.assert_allclose(float(loss), float(fx["ddqn/loss"]), rtol=RTOL)
    _assert_module(agent.q_network_local, fx, "ddqn/local1", "DDQN local net after one update")
    _assert_module(agent.q_network_target, fx, "ddqn/target1", "DDQN target net after the soft update")


def _sac(device):
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.SAC_Discrete import SAC_Discrete
    fx = np.load(FIX)
    lr, gamma, clip, tau, log_alpha0 = fx["sac/hyper"]
    agent = SAC_Discrete(_Env(device), hidden_size=16, hidden_layer=2, seed=1,
                         hyper={"learning_rate": float(lr), "discount_rate": float(gamma), "gradient_clipping_norm": float(clip),
                                "tau": float(tau), "batch_size": 24, "buffer_size": 64,
                                "automatically_tune_entropy_hyper_parameter": True})
    nets = {"critic1": agent.critic_local, "critic2": agent.critic_local_2, "target1": agent.critic_target,
            "target2": agent.critic_target_2, "actor": agent.actor_local}
    for name, net in nets.items():
        _load(net, fx, "sac/%s0" % name, device)
    with torch.no_grad():
        agent.log_alpha.fill_(float(log_alpha0))
    agent.alpha = agent.log_alpha.exp()
    agent.memory.add_batch(_t(fx, "sac/states", device, torch.float64), _t(fx, "sac/actions", device), _t(fx, "sac/rewards", device, torch.float64),
                           _t(fx, "sac/next_states", device, torch.float64), _t(fx, "sac/dones", device, torch.uint8))
    q1, q2, pl = (float(v) for v in agent.learn())                                          # SAC_Discrete.py:293-352
    np.testing.assert_allclose([q1, q2, pl], fx["sac/losses"][:3], rtol=RTOL)
    np.testing.assert_allclose(agent.log_alpha.detach().cpu().numpy(), fx["sac/log_alpha1"], rtol=RTOL)
    for name, net in nets.items():
        _assert_module(net, fx, "sac/%s1" % name, "SAC %s after one learn()" % name)


def _a3c(device):
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import DA3C
    fx = np.load(FIX)
    lr, gamma, clip = fx["a3c/hyper"]
    agent = DA3C(None, _Env(device), reward_policy=0, hidden_size=16, hidden_layer=2, seed=1,
                 hyper={"learning_rate": float(lr), "discount_rate": float(gamma), "gradient_clipping_norm": float(clip)})
    nets = {"task": agent.actor_task_model, "machine": agent.actor_machine_model, "critic": agent.critic_model}
    for name, net in nets.items():
        _load(net, fx, "a3c/%s0" % name, device)
    states = _t(fx, "a3c/states", device).unsqueeze(1)                                      # [T, 1, 30]: one worker's episode
    actions = torch.from_numpy(fx["a3c/actions"]).to(device).unsqueeze(1)
    rewards = _t(fx, "a3c/rewards", device, torch.float64).unsqueeze(1)
    valid = torch.ones_like(rewards)
    losses = agent.learn_from_rollout(states, actions, rewards, valid)                      # A3C_v5.1.py:363-437
    np.testing.assert_allclose(losses, fx["a3c/losses"], rtol=5e-5)
    for name, net in nets.items():
        _assert_module(net, fx, "a3c/%s1" % name, "A3C %s net after one worker update" % name)


def _ppo(device):
    """PPO.run_one_policy_network's learning round (MPPPO.py:254-270,301-370) by the reference's methods, as shipped: the
    critic receives no gradient there (:319), so the recorded critic is unchanged -- PPOLearner(train_critic=False)."""
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import PPOLearner, discounted_returns, normalise_returns
    fx = np.load(os.path.join(os.path.dirname(FIX), "ppo_update.npz"))
    lr, gamma, clip_eps, clip_norm, iters = fx["ppo/hyper"]
    L = PPOLearner(25, 18, 16, 2, 2, device=device, seed=3, train_critic=False,
                   hyper={"learning_rate": float(lr), "discount_rate": float(gamma), "clip_epsilon": float(clip_eps),
                          "gradient_clipping_norm": float(clip_norm), "learning_iterations_per_round_critic": int(iters),
                          "learning_iterations_per_round_actor": int(iters)})
    _load(L.actor_new, fx, "ppo/actor0", device); _load(L.actor_old, fx, "ppo/actor0", device); _load(L.critic, fx, "ppo/critic0", device)
    states, rewards, old_lp = _t(fx, "ppo/states", device), _t(fx, "ppo/rewards", device), _t(fx, "ppo/old_log_prob", device)
    actions = torch.from_numpy(fx["ppo/actions"]).to(device)
    T = states.shape[0]
    valid = torch.ones(T, 1, device=device)
    G = discounted_returns(rewards[:, None], valid, float(gamma))                                   # :301-312
    np.testing.assert_array_equal(G[:, 0].cpu().numpy(), fx["ppo/returns_raw"])                     # same f32 recurrence: same bits
    Gn = normalise_returns(G, valid)                                                                # :258-261
    np.testing.assert_allclose(Gn[:, 0].cpu().numpy(), fx["ppo/returns"], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        adv = Gn[:, 0] - L.critic(states).squeeze(1)                                                # :263
    np.testing.assert_allclose(adv.cpu().numpy(), fx["ppo/advantages"], rtol=1e-5, atol=2e-6)
    L.learn(states[:, None, :], actions[:, None], old_lp[:, None], Gn, valid)                       # :314-323
    _assert_module(L.actor_new, fx, "ppo/actor1", "PPO actor after one learning round")
    _assert_module(L.critic, fx, "ppo/critic1", "PPO critic (no gradient reaches it in the reference as shipped)")
    _assert_module(L.actor_old, fx, "ppo/actor1", "PPO old policy after equalise_policies")


@pytest.mark.parametrize("case", [_ddqn, _sac, _a3c, _ppo], ids=["ddqn", "sac", "a3c", "ppo"])
def test_update_matches_the_reference_classes_on_cpu(case):
    torch.set_num_threads(1)
    case("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [_ddqn, _sac, _a3c, _ppo], ids=["ddqn", "sac", "a3c", "ppo"])
def test_update_matches_the_reference_classes_on_gpu(case):
    assert torch.cuda.is_available()
    case("cuda:0")
