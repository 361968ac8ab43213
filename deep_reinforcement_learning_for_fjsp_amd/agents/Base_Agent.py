"""Agent-side surface the reference's training scripts are written against
(agents/Base_Agent.py:9-137; SURVEY.md 8b "agent-side surface to keep").

Only the interface is shared with the reference -- method names, argument meaning
and observable behaviour -- so that `class PPO(Base_Agent, Config)`-style agents
subclass it unchanged.  The bodies are written for the batched device path: updates
run under `torch.no_grad()` on whole parameter lists, episode traces are kept in one
table, and nothing here assumes a single environment.
"""
import torch

_EPISODE_FIELDS = ("state", "action", "reward", "next_state", "done")
_TRACE_NAMES = {"state": "episode_states", "action": "episode_actions", "reward": "episode_rewards",
                "next_state": "episode_next_states", "done": "episode_dones"}


def _parameter_pairs(source, destination):
    return zip(source.parameters(), destination.parameters())


class Base_Agent(object):
    """Common state and helpers of every agent (reference :10-31)."""

    def __init__(self):
        self.state_size = self.action_size = None
        self.episode_number = 0
        self.turn_off_exploration = False
        self.device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")   # :14
        self.reset_game()

    # -- methods an agent must provide --------------------------------------------------------
    def step(self):
        raise ValueError("step方法需要在对应智能体中重写")                     # :33

    def get_state_size(self):
        # the reference RETURNS the exception object here instead of raising it (:39); kept, callers may rely on it
        return ValueError("该方法需要在对应智能体中重写")

    # -- per-episode bookkeeping (:42-61) ------------------------------------------------------
    def reset_game(self):
        for field in _EPISODE_FIELDS:
            setattr(self, field, False if field == "done" else None)
            setattr(self, _TRACE_NAMES[field], [])

    def track_episodes_data(self):
        for field in _EPISODE_FIELDS:
            getattr(self, _TRACE_NAMES[field]).append(getattr(self, field))

    # -- replay memory (:63-71) ------------------------------------------------------------------
    def enough_experiences_to_learn_from(self, memory, batch_size):
        """Strictly more experiences than one batch."""
        return len(memory) > batch_size

    def save_experience(self, memory=None, experience=None):
        """Append `experience` (default: the agent's current transition) to `memory`."""
        row = experience if experience is not None else tuple(getattr(self, f) for f in _EPISODE_FIELDS)
        memory.add_experience(*row)

    # -- optimisation (:73-87) -------------------------------------------------------------------
    def take_optimisation_step(self, optimizer, network, loss, clipping_norm=None, retain_graph=False):
        """zero_grad -> backward -> optional gradient-norm clip of every given network -> step."""
        optimizer.zero_grad()
        loss.backward(retain_graph=retain_graph)
        if clipping_norm is not None:
            for net in (network if isinstance(network, list) else [network]):
                torch.nn.utils.clip_grad_norm_(net.parameters(), clipping_norm)
        optimizer.step()

    @torch.no_grad()
    def soft_update_of_target_network(self, local_model, target_model, tau=0.005):
        """target <- tau * local + (1 - tau) * target, parameter by parameter."""
        for local_param, target_param in _parameter_pairs(local_model, target_model):
            target_param.copy_(tau * local_param + (1.0 - tau) * target_param)

    # -- exploration switches (:89-97) ------------------------------------------------------------
    def turn_on_any_epsilon_greedy_exploration(self):
        self.turn_off_exploration = False

    def turn_off_any_epsilon_greedy_exploration(self):
        self.turn_off_exploration = True

    # -- layer freezing (:99-113) -------------------------------------------------------------------
    def freeze_all_but_output_layers(self, network):
        for name, param in network.named_parameters():
            if not any(tag in name for tag in ("hidden", "output", "embedding")):
                raise AssertionError("Name {} of network layers not understood".format(name))
            param.requires_grad = "output" in name and param.requires_grad

    def unfreeze_all_layers(self, network):
        network.requires_grad_(True)

    # -- model plumbing (:115-137) --------------------------------------------------------------------
    @staticmethod
    def move_gradients_one_model_to_another(from_model, to_model, set_from_gradients_to_zero=False):
        for src, dst in _parameter_pairs(from_model, to_model):
            dst._grad = src.grad.clone()
            if set_from_gradients_to_zero:
                src._grad = None

    @staticmethod
    def copy_model_over(from_model, to_model):
        """Exact copy (the two models have identical parameter sets)."""
        to_model.load_state_dict(from_model.state_dict())

    @staticmethod
    def copy_model_over_dict(from_model, to_model):
        """Copy the entries `from_model` has into `to_model`'s state, keeping the rest."""
        merged = dict(to_model.state_dict())
        merged.update(from_model.state_dict())
        to_model.load_state_dict(merged)
