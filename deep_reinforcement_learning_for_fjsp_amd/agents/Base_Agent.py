"""Mirror of the reference's agents/Base_Agent.py:9-137 (the interface the
training scripts are written against; SURVEY.md 8b "agent-side surface to keep").
Same method names, argument meaning and behaviour; PyTorch-ROCm underneath."""
import torch


class Base_Agent(object):
    def __init__(self):
        self.action_size = None
        self.state_size = None
        self.episode_number = 0
        self.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")   # Base_Agent.py:14
        self.turn_off_exploration = False
        self.reset_game()

    def step(self):
        raise ValueError("step方法需要在对应智能体中重写")                               # :33

    def get_state_size(self):
        return ValueError("该方法需要在对应智能体中重写")                                 # :39 (returns, not raises)

    def reset_game(self):                                                            # :42-53
        self.state = None
        self.next_state = None
        self.action = None
        self.reward = None
        self.done = False
        self.episode_states = []
        self.episode_rewards = []
        self.episode_actions = []
        self.episode_next_states = []
        self.episode_dones = []

    def track_episodes_data(self):                                                   # :55-61
        self.episode_states.append(self.state)
        self.episode_actions.append(self.action)
        self.episode_rewards.append(self.reward)
        self.episode_next_states.append(self.next_state)
        self.episode_dones.append(self.done)

    def enough_experiences_to_learn_from(self, memory, batch_size):                  # :63-65
        return len(memory) > batch_size

    def save_experience(self, memory=None, experience=None):                         # :67-71
        if experience is None:
            experience = self.state, self.action, self.reward, self.next_state, self.done
        memory.add_experience(*experience)

    def take_optimisation_step(self, optimizer, network, loss, clipping_norm=None, retain_graph=False):  # :73-82
        if not isinstance(network, list):
            network = [network]
        optimizer.zero_grad()
        loss.backward(retain_graph=retain_graph)
        if clipping_norm is not None:
            for net in network:
                torch.nn.utils.clip_grad_norm_(net.parameters(), clipping_norm)
        optimizer.step()

    def soft_update_of_target_network(self, local_model, target_model, tau=0.005):   # :84-87
        for target_param, local_param in zip(target_model.parameters(), local_model.parameters()):
            target_param.data.copy_(tau * local_param.data + (1.0 - tau) * target_param.data)

    def turn_on_any_epsilon_greedy_exploration(self):
        self.turn_off_exploration = False

    def turn_off_any_epsilon_greedy_exploration(self):
        self.turn_off_exploration = True

    def freeze_all_but_output_layers(self, network):                                 # :99-107
        for name, param in network.named_parameters():
            assert "hidden" in name or "output" in name or "embedding" in name, \
                "Name {} of network layers not understood".format(name)
            if "output" not in name:
                param.requires_grad = False

    def unfreeze_all_layers(self, network):
        for param in network.parameters():
            param.requires_grad = True

    @staticmethod
    def move_gradients_one_model_to_another(from_model, to_model, set_from_gradients_to_zero=False):   # :115-121
        for src, dst in zip(from_model.parameters(), to_model.parameters()):
            dst._grad = src.grad.clone()
            if set_from_gradients_to_zero:
                src._grad = None

    @staticmethod
    def copy_model_over(from_model, to_model):                                       # :123-126
        to_model.load_state_dict(from_model.state_dict())

    @staticmethod
    def copy_model_over_dict(from_model, to_model):                                  # :128-137
        sd = to_model.state_dict()
        sd.update(from_model.state_dict())
        to_model.load_state_dict(sd)
