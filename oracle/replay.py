"""Oracle: replay ring buffer (TEST INFRASTRUCTURE).

Restates reference ``src/replaybuffer.py:5-63``.
"""
import numpy as np


class RefReplayBuffer:
    def __init__(self, buffer_capacity=100000, batch_size=64, num_states=None, num_actions=None, dtype=np.float64):
        self.buffer_capacity, self.batch_size = buffer_capacity, batch_size
        self.buffer_counter = 0
        self.state_buffer = np.zeros((buffer_capacity, num_states), dtype=dtype)  # :31-34
        self.action_buffer = np.zeros((buffer_capacity, num_actions), dtype=dtype)
        self.reward_buffer = np.zeros((buffer_capacity, 1), dtype=dtype)
        self.next_state_buffer = np.zeros((buffer_capacity, num_states), dtype=dtype)

    def add(self, obs_tuple):  # :37-47
        index = self.buffer_counter % self.buffer_capacity
        self.state_buffer[index] = obs_tuple[0]
        self.action_buffer[index] = obs_tuple[1]
        self.reward_buffer[index] = obs_tuple[2]
        self.next_state_buffer[index] = obs_tuple[3]
        self.buffer_counter += 1

    def sample_indices(self):  # :52-54 -- with replacement, global legacy RNG
        record_range = min(self.buffer_counter, self.buffer_capacity)
        return np.random.choice(record_range, self.batch_size)

    def gather(self, idx):  # :57-61 (reward cast to f32)
        return (self.state_buffer[idx], self.action_buffer[idx], self.reward_buffer[idx].astype(np.float32),
                self.next_state_buffer[idx])

    def sample(self):
        return self.gather(self.sample_indices())


def ring_index(counter, capacity):
    """Write slot of the next add (replaybuffer.py:40) -- integer, bit-exact."""
    return counter % capacity


def sample_range(counter, capacity):
    """Exclusive upper bound of sampled indices (replaybuffer.py:52)."""
    return min(counter, capacity)
