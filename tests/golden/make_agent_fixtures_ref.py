#!/usr/bin/env python3
"""tests/golden/agent_updates.npz from the REFERENCE'S OWN agent classes (build container only: needs /root/reference).

Round 2 pinned the off-policy / actor-critic updates with a hand transcription (make_agent_fixtures.py).  This script
imports the reference modules themselves

    agents/DDQN/DDQN.py            DDQN.learn / compute_loss                                   :168-209
    agents/HMPSAC/SAC_Discrete.py  SAC_Discrete.calculate_critic_losses / update_critic_parameters /
                                   calculate_actor_loss / calculate_entropy_tuning_loss / update_actor_parameters  :293-352
    agents/HMPSAC/A3C_v5.1.py      Actor_Critic_Worker.calculate_total_loss / put_gradients_in_queue and the three
                                   optimiser steps of DA3C.update_shared_model (utilities.Utility_Functions.SharedAdam)   :164-187,363-437
    agents/MPPPO/MPPPO.py          PPO.calculate_discounted_returns / critic_actor_learn / calculate_all_ratio_of_policy_probabilities /
                                   calculate_actor_loss / take_policy_new_optimisation_step   :301-370 (a second file: ppo_update.npz)
    agents/Base_Agent.py           take_optimisation_step, soft_update_of_target_network       :73-87

and calls THOSE methods on objects built without running the constructors' environment / checkpoint / visdom code
(`object.__new__` + the attributes the methods read), at the same fixed weights and batches as before.  What keeps the
modules from importing in this image is stubbed in sys.modules: `visdom` (module-level `Visdom()` + `vis.line`),
`nn_builder.pytorch.NN` (utilities/Utility_Functions.py:6), `openpyxl` / `docplex` (oracle/ref_shim), and
`utilities.Utility_Class.AddData.add_data` (appends to a `D:/...` csv at import).  Nothing of the reference travels:
only the arrays in the .npz do.

The transcription stays beside it as a cross-check: --compare prints the largest difference per array between the
two generators (a transcription error shows up there).

    python tests/golden/make_agent_fixtures_ref.py [--compare]
"""
import importlib.util
import os
import queue
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def bootstrap_reference():
    """Make the reference's agent modules importable here; returns (ddqn_module, sac_module, a3c_module)."""
    if not os.path.isdir(REF):
        raise SystemExit("needs %s (build container only)" % REF)
    for p in (os.path.join(REPO, "oracle", "ref_shim"), REF, os.path.join(REF, "agents", "MPPPO")):
        if p not in sys.path:
            sys.path.insert(0, p)

    class _Anything(object):                      # Visdom(), vis.line(...), NN(...): accepted and ignored
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    vis = types.ModuleType("visdom"); vis.Visdom = _Anything
    nb, nbp, nbn = types.ModuleType("nn_builder"), types.ModuleType("nn_builder.pytorch"), types.ModuleType("nn_builder.pytorch.NN")
    nbn.NN = _Anything
    sys.modules.update({"visdom": vis, "nn_builder": nb, "nn_builder.pytorch": nbp, "nn_builder.pytorch.NN": nbn})
    import matplotlib
    matplotlib.use("Agg")
    import utilities.Utility_Class as UC
    UC.AddData.add_data = lambda self, data: None          # (module level: add_data_object.add_data([...]) into D:/...)
    import agents.DDQN.DDQN as ref_ddqn
    import agents.HMPSAC.SAC_Discrete as ref_sac
    spec = importlib.util.spec_from_file_location("ref_a3c_v5_1", os.path.join(REF, "agents", "HMPSAC", "A3C_v5.1.py"))
    ref_a3c = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_a3c)
    return ref_ddqn, ref_sac, ref_a3c


def import_reference_mpppo():
    import agents.MPPPO.MPPPO as ref_mpppo          # (after bootstrap_reference(): same stubs; `from Buffer import ...` via sys.path)
    return ref_mpppo


def dump(prefix, module, out, rename=None):
    for k, v in module.state_dict().items():
        if rename:
            k = k.replace(*rename)
        out["%s/%s" % (prefix, k)] = v.detach().cpu().numpy().copy()


def load_as(module, donor):
    """Give a reference network the weights the seeded stand-in of the same architecture drew (the fixture's inputs stay
    what they were); parameter ORDER is the architecture's, names may differ (layers_1 / layers_2 in A3C_v5.1.py)."""
    sd = module.state_dict()
    vals = list(donor.state_dict().values())
    assert len(sd) == len(vals)
    module.load_state_dict({k: v.clone() for k, v in zip(sd.keys(), vals)})


# ------------------------------------------------------------------ DDQN
def ddqn_fixture(out, ref_ddqn, T):
    torch.manual_seed(11)
    hp = {"learning_rate": 1e-3, "discount_rate": 1.0, "gradient_clipping_norm": 5.0, "tau": 0.005}
    d_local, d_target = T.DDQNActorNet(18, 16, 2, 20), T.DDQNActorNet(18, 16, 2, 20)        # same seeded draws as the transcription
    B = 32
    states, next_states = torch.randn(B, 18), torch.randn(B, 18)
    actions = torch.randint(0, 20, (B, 1)).float()
    rewards = -torch.rand(B, 1)
    dones = (torch.rand(B, 1) < 0.25).float()
    agent = object.__new__(ref_ddqn.DDQN)                                                    # (no __init__: it builds environments from D:/)
    agent.hyper_parameters = dict(hp)
    agent.q_network_local, agent.q_network_target = ref_ddqn.ActorNet(18, 16, 2, 20), ref_ddqn.ActorNet(18, 16, 2, 20)
    load_as(agent.q_network_local, d_local); load_as(agent.q_network_target, d_target)
    agent.q_network_optimizer = torch.optim.Adam(agent.q_network_local.parameters(), lr=hp["learning_rate"], eps=1e-4)   # DDQN.py:86
    dump("ddqn/local0", agent.q_network_local, out); dump("ddqn/target0", agent.q_network_target, out)
    for k, v in (("states", states), ("next_states", next_states), ("actions", actions), ("rewards", rewards), ("dones", dones)):
        out["ddqn/" + k] = v.numpy().copy()
    out["ddqn/hyper"] = np.array([hp["learning_rate"], hp["discount_rate"], hp["gradient_clipping_norm"], hp["tau"]])
    loss = agent.compute_loss(states, next_states, rewards, actions, dones)                   # :182-188 (train mode: batch statistics)
    out["ddqn/loss"] = np.array(float(loss.detach()))
    # compute_loss ran the BatchNorm layers in train mode once: restore the running statistics so that learn() starts from the
    # recorded state, like one call of the reference's learn() does
    load_as(agent.q_network_local, d_local); load_as(agent.q_network_target, d_target)
    agent.learn(experiences=(states, actions, rewards, next_states, dones))                   # :168-180
    dump("ddqn/local1", agent.q_network_local, out); dump("ddqn/target1", agent.q_network_target, out)


# ------------------------------------------------------------------ SAC-discrete
def sac_fixture(out, ref_sac, T):
    hp = {"learning_rate": 3e-3, "discount_rate": 0.99, "gradient_clipping_norm": 5.0, "tau": 0.005}
    agent = object.__new__(ref_sac.SAC_Discrete)
    agent.hyper_parameters = dict(hp)
    agent.action_types, agent.action_size = "DISCRETE", 3
    agent.critic_local, agent.critic_local_2 = ref_sac.CriticNet(30, 16, 2, 3), ref_sac.CriticNet(30, 16, 2, 3)
    agent.critic_target, agent.critic_target_2 = ref_sac.CriticNet(30, 16, 2, 3), ref_sac.CriticNet(30, 16, 2, 3)
    agent.actor_local = ref_sac.PolicyNet(30, 16, 2, 3)
    torch.manual_seed(12)                    # (after the reference networks' own initialisation draws: the recorded inputs stay
    mk = lambda sm: T.ReluStack(30, 16, 2, 3, sm)     # what the round-2 fixtures held)
    donors = [mk(False), mk(False), mk(False), mk(False), mk(True)]
    B = 24
    nets = (("critic1", agent.critic_local), ("critic2", agent.critic_local_2), ("target1", agent.critic_target),
            ("target2", agent.critic_target_2), ("actor", agent.actor_local))
    for (name, net), donor in zip(nets, donors):
        load_as(net, donor)
    agent.critic_optimizer = torch.optim.Adam(agent.critic_local.parameters(), lr=hp["learning_rate"], eps=1e-4)          # :159-160,167
    agent.critic_optimizer_2 = torch.optim.Adam(agent.critic_local_2.parameters(), lr=hp["learning_rate"], eps=1e-4)
    agent.actor_optimizer = torch.optim.Adam(agent.actor_local.parameters(), lr=hp["learning_rate"], eps=1e-4)
    agent.automatic_entropy_tuning = True
    agent.target_entropy = -np.log((1.0 / 3)) * 0.98                                                                       # :172
    agent.log_alpha = torch.tensor([0.3], requires_grad=True)
    agent.alpha = agent.log_alpha.exp()
    agent.alpha_optim = torch.optim.Adam([agent.log_alpha], lr=hp["learning_rate"], eps=1e-4)
    state_batch, next_state_batch = torch.randn(B, 30), torch.randn(B, 30)
    action_batch = torch.randint(0, 3, (B, 1)).float()
    reward_batch = -torch.rand(B, 1)
    done_batch = (torch.rand(B, 1) < 0.2).float()
    for name, net in nets:
        dump("sac/%s0" % name, net, out)
    for k, v in (("states", state_batch), ("next_states", next_state_batch), ("actions", action_batch), ("rewards", reward_batch),
                 ("dones", done_batch)):
        out["sac/" + k] = v.numpy().copy()
    out["sac/hyper"] = np.array([hp["learning_rate"], hp["discount_rate"], hp["gradient_clipping_norm"], hp["tau"], 0.3])
    # learn() :293-302 with the batch handed in instead of sampled from the replay memory
    qf1_loss, qf2_loss = agent.calculate_critic_losses(state_batch, action_batch, reward_batch, next_state_batch, done_batch)
    agent.update_critic_parameters(qf1_loss, qf2_loss)
    policy_loss, log_pi = agent.calculate_actor_loss(state_batch)
    alpha_loss = agent.calculate_entropy_tuning_loss(log_pi)
    agent.update_actor_parameters(policy_loss, alpha_loss)
    out["sac/losses"] = np.array([float(qf1_loss.detach()), float(qf2_loss.detach()), float(policy_loss.detach()), float(alpha_loss.detach())])
    out["sac/log_alpha1"] = agent.log_alpha.detach().numpy().copy()
    for name, net in nets:
        dump("sac/%s1" % name, net, out)


# ------------------------------------------------------------------ one A3C worker episode + the shared optimiser step
def a3c_fixture(out, ref_a3c, T):
    shared = (ref_a3c.TaskPolicyNet(30, 16, 2, 12), ref_a3c.MachinePolicyNet(31, 16, 2, 10), ref_a3c.CriticNet(30, 16, 2, 1))
    torch.manual_seed(13)
    rs = np.random.RandomState(13)
    lr, discount_rate, gradient_clipping_norm = 1e-3, 0.99, 5.0
    donors = (T.ReluStack(30, 16, 2, 12, True), T.ReluStack(31, 16, 2, 10, True), T.ReluStack(30, 16, 2, 1, False))
    for net, donor in zip(shared, donors):
        load_as(net, donor)
    import copy
    from utilities.Utility_Functions import SharedAdam, create_actor_distribution
    Tn = 14
    episode_states = [rs.randn(30) for _ in range(Tn)]
    episode_actions = [np.array([rs.randint(0, 12), rs.randint(0, 10)]) for _ in range(Tn)]
    episode_rewards = [-float(rs.randint(0, 40)) for _ in range(Tn)]
    for name, net, ren in (("task", shared[0], ("layers_1.", "layers.")), ("machine", shared[1], ("layers_2.", "layers.")),
                           ("critic", shared[2], None)):
        dump("a3c/%s0" % name, net, out, ren)
    out["a3c/states"] = np.stack(episode_states)
    out["a3c/actions"] = np.stack(episode_actions).astype(np.int64)
    out["a3c/rewards"] = np.array(episode_rewards)
    out["a3c/hyper"] = np.array([lr, discount_rate, gradient_clipping_norm])
    # the master (DA3C) with its SharedAdam optimisers :112-114, and one worker with local copies :155-160
    master = object.__new__(ref_a3c.DA3C)
    master.actor_task_model, master.actor_machine_model, master.critic_model = shared
    master.actor_task_optimizer = SharedAdam(shared[0].parameters(), lr=lr, eps=1e-4)
    master.actor_machine_optimizer = SharedAdam(shared[1].parameters(), lr=lr, eps=1e-4)
    master.critic_optimizer = SharedAdam(shared[2].parameters(), lr=lr, eps=1e-4)
    worker = object.__new__(ref_a3c.Actor_Critic_Worker)
    worker.action_types = "DISCRETE"
    worker.discount_rate, worker.normalise_rewards, worker.gradient_clipping_norm = discount_rate, True, gradient_clipping_norm
    worker.local_actor_task_model, worker.local_actor_machine_model, worker.local_critic_model = (copy.deepcopy(m) for m in shared)
    worker.local_actor_task_optimizer = torch.optim.Adam(worker.local_actor_task_model.parameters(), lr=0.0, eps=1e-4)       # :229-231 (zero_grad only)
    worker.local_actor_machine_optimizer = torch.optim.Adam(worker.local_actor_machine_model.parameters(), lr=0.0, eps=1e-4)
    worker.local_critic_optimizer = torch.optim.Adam(worker.local_critic_model.parameters(), lr=0.0, eps=1e-4)
    q_task, q_machine, q_critic = queue.Queue(), queue.Queue(), queue.Queue()
    worker.gradient_updates_queue_actor_task, worker.gradient_updates_queue_actor_machine, worker.gradient_updates_queue_critic = q_task, q_machine, q_critic
    # the episode loop :270-283 with the recorded actions in place of the sampled ones (the worker's own forward calls)
    worker.episode_states, worker.episode_rewards = episode_states, episode_rewards
    worker.episode_log_action_task_probabilities, worker.episode_log_action_machine_probabilities, worker.critic_outputs = [], [], []
    for state, actions in zip(episode_states, episode_actions):
        s = torch.from_numpy(state).float().unsqueeze(0)                                                # :319
        dist_task = create_actor_distribution("DISCRETE", worker.local_actor_task_model.forward(s), 12)
        worker.episode_log_action_task_probabilities.append(
            worker.calculate_log_action_probability(int(actions[0]), dist_task))                          # :336,357-361 (a scalar action)
        state_add = np.append(state, actions[0])                                                        # :271
        s2 = torch.from_numpy(state_add).float().unsqueeze(0)
        dist_machine = create_actor_distribution("DISCRETE", worker.local_actor_machine_model.forward(s2), 10)
        worker.episode_log_action_machine_probabilities.append(
            worker.calculate_log_action_probability(int(actions[1]), dist_machine))
        worker.critic_outputs.append(worker.get_critic_value(worker.local_critic_model, state))         # :350-355
    returns = worker.normalise_discounted_returns(np.array(worker.calculate_discounted_returns()))      # recorded; recomputed inside below
    critic_loss, actor_task_loss, actor_machine_loss = worker.calculate_total_loss()                    # :363-372
    worker.put_gradients_in_queue(critic_loss, actor_task_loss, actor_machine_loss)                     # :419-437
    # DA3C.update_shared_model :164-187, one turn of its loop (the queues are read once)
    g_task, g_machine, g_critic = q_task.get(), q_machine.get(), q_critic.get()
    for opt, grads, model in ((master.actor_task_optimizer, g_task, shared[0]), (master.actor_machine_optimizer, g_machine, shared[1]),
                              (master.critic_optimizer, g_critic, shared[2])):
        opt.zero_grad()
        for grad, params in zip(grads, model.parameters()):
            params._grad = grad
        opt.step()
    out["a3c/returns"] = np.asarray(returns, np.float64).copy()
    out["a3c/losses"] = np.array([float(critic_loss.detach()), float(actor_task_loss.detach()), float(actor_machine_loss.detach())])
    for name, net, ren in (("task", shared[0], ("layers_1.", "layers.")), ("machine", shared[1], ("layers_2.", "layers.")),
                           ("critic", shared[2], None)):
        dump("a3c/%s1" % name, net, out, ren)


# ------------------------------------------------------------------ one clipped-PPO learning round (MPPPO.py:254-270,301-370)
def ppo_fixture(out, ref_mpppo):
    """The learning round of PPO.run_one_policy_network (:254-270) on a recorded episode, by the reference's own methods.
    What cannot be used as shipped, and what the product does instead (both documented since round 1):
      * :319 `critic_loss = critic_loss.clone().detach().requires_grad_(True)` cuts the critic out of the graph: the
        reference's critic never receives a gradient.  critic_actor_learn() is called as shipped -- the recorded critic is
        therefore UNCHANGED -- and the product's PPOLearner(train_critic=False) reproduces exactly that; its default
        (train_critic=True) is the documented fix;
      * :375 equalise_policies() reads `old_param.algorithm_means` (AttributeError): not called; the product copies the
        parameters (the evident intent)."""
    torch.manual_seed(14)
    S, A, H, T = 25, 18, 16, 40
    hp = {"clip_epsilon": 0.2, "gradient_clipping_norm": 5.0, "normalized_rewards": True, "standardized_rewards": True}
    agent = object.__new__(ref_mpppo.PPO)
    agent.device = torch.device("cpu")
    agent.hyper_parameters = dict(hp)
    agent.action_types, agent.action_size = "DISCRETE", A
    agent.discount_rate = 0.99
    agent.learning_iterations_per_round_critic = 3
    agent.actor_new, agent.actor_old = ref_mpppo.ActorNet(S, H, 2, A), ref_mpppo.ActorNet(S, H, 2, A)
    agent.actor_old.load_state_dict(agent.actor_new.state_dict())
    agent.critic_local = ref_mpppo.CriticNet(S, H, 2, 1)
    agent.actor_optimizer = torch.optim.Adam(agent.actor_new.parameters(), lr=1e-3, eps=1e-4)             # :144
    agent.critic_optimizer = torch.optim.Adam(agent.critic_local.parameters(), lr=1e-3, eps=1e-4)         # :146
    states = torch.randn(T, S)
    with torch.no_grad():
        dist = torch.distributions.Categorical(agent.actor_new(states))
        actions = dist.sample()
        # an exploration override now and then, like :279-281: the old log-probability is the taken action's
        actions = torch.where(torch.rand(T) < 0.2, torch.randint(0, A, (T,)), actions)
        old_log_prob = dist.log_prob(actions)
    rewards = -torch.rand(T) * 50.0
    dump("ppo/actor0", agent.actor_new, out); dump("ppo/critic0", agent.critic_local, out)
    out["ppo/states"], out["ppo/actions"], out["ppo/rewards"] = states.numpy().copy(), actions.numpy().astype(np.int64), rewards.numpy().copy()
    out["ppo/old_log_prob"] = old_log_prob.numpy().copy()
    out["ppo/hyper"] = np.array([1e-3, agent.discount_rate, hp["clip_epsilon"], hp["gradient_clipping_norm"], 3.0])
    # :254-263 with the replay memory's sample() replaced by the recorded episode (Buffer.py:40-48: f32 tensors)
    agent.episode_states, agent.episode_actions, agent.episode_rewards = states, actions.float(), rewards
    agent.actor_old_log_prob = old_log_prob.float().detach()
    agent.discounted_returns = agent.calculate_discounted_returns()                                         # :301-312
    out["ppo/returns_raw"] = agent.discounted_returns.numpy().copy()
    if agent.hyper_parameters["normalized_rewards"]:                                                        # :258-259
        agent.discounted_returns = (agent.discounted_returns - agent.discounted_returns.min()) / (agent.discounted_returns.max() - agent.discounted_returns.min() + 1e-8)
    if agent.hyper_parameters["standardized_rewards"]:                                                      # :260-261
        agent.discounted_returns = (agent.discounted_returns - agent.discounted_returns.mean()) / (agent.discounted_returns.std() + 1e-8)
    agent.critic_targets = agent.discounted_returns                                                         # :262
    agent.advantages = agent.discounted_returns - agent.critic_local(agent.episode_states).squeeze(1).detach()   # :263
    out["ppo/returns"] = agent.discounted_returns.numpy().copy()
    out["ppo/advantages"] = agent.advantages.numpy().copy()
    with torch.no_grad():
        ratio0 = agent.calculate_all_ratio_of_policy_probabilities(agent.actor_old_log_prob)
        out["ppo/actor_loss0"] = np.array(float(agent.calculate_actor_loss([ratio0], agent.advantages)))
    agent.critic_actor_learn()                                                                              # :314-323
    dump("ppo/actor1", agent.actor_new, out); dump("ppo/critic1", agent.critic_local, out)


def main():
    torch.set_num_threads(1)
    sys.path.insert(0, HERE)
    import make_agent_fixtures as T                 # the transcription: donor networks (same seeded draws) and the cross-check
    ref_ddqn, ref_sac, ref_a3c = bootstrap_reference()
    out = {}
    ddqn_fixture(out, ref_ddqn, T)
    sac_fixture(out, ref_sac, T)
    a3c_fixture(out, ref_a3c, T)
    if "--compare" in sys.argv:
        old = {}
        T.ddqn_fixture(old); T.sac_fixture(old); T.a3c_fixture(old)
        assert sorted(old) == sorted(out), sorted(set(old) ^ set(out))
        worst = {}
        for k in sorted(out):
            d = float(np.max(np.abs(np.asarray(out[k], np.float64) - np.asarray(old[k], np.float64)))) if out[k].size else 0.0
            grp = k.split("/")[0] + "/" + k.split("/")[1]
            worst[grp] = max(worst.get(grp, 0.0), d)
        for g, d in sorted(worst.items()):
            print("%-22s max |reference - transcription| = %.3e" % (g, d))
    ppo = {}
    ppo_fixture(ppo, import_reference_mpppo())
    ppath = os.path.join(HERE, "ppo_update.npz")
    np.savez_compressed(ppath, **ppo)
    print("%s: %d arrays, %.1f KB (PPO.critic_actor_learn of the reference); actor loss before the round %.8f"
          % (ppath, len(ppo), os.path.getsize(ppath) / 1024, ppo["ppo/actor_loss0"]))
    path = os.path.join(HERE, "agent_updates.npz")
    np.savez_compressed(path, **out)
    print("%s: %d arrays, %.1f KB, generated by the reference's classes; torch %s" % (path, len(out), os.path.getsize(path) / 1024, torch.__version__))
    print("ddqn loss %.8f | sac losses %s | a3c losses %s" % (out["ddqn/loss"], out["sac/losses"], out["a3c/losses"]))


if __name__ == "__main__":
    main()
