"""Block-config parser and schedulers (API mirror of latentfusion/utils.py:33-54,125-162)."""
import math
from bisect import bisect_right


def parse_block_str(s):
    return s if s in {'I', 'U', 'D'} else int(s)


def parse_block_config(s, delimiter=',', group_delimiter=':'):
    """'64,D,128:128,U,64' -> [[64,'D',128],[128,'U',64]]; 'none'/'' -> []."""
    if s.lower() == 'none' or len(s) == 0:
        return []

    def blocks(part):
        return [parse_block_str(tok) for tok in part.split(delimiter)] if part else []
    if group_delimiter in s:
        return [blocks(part) for part in s.split(group_delimiter)]
    return blocks(s)


class MultiStepMilestoneScheduler:
    def __init__(self, initial_value, milestones, gamma):
        self.initial_value, self.milestones, self.gamma = initial_value, milestones, gamma

    def get(self, step):
        if self.milestones is None:
            return self.initial_value
        return self.initial_value * self.gamma ** bisect_right(self.milestones, step)


class LinearScheduler:
    def __init__(self, initial_value, end_value, num_steps):
        self.initial_value, self.end_value, self.num_steps = initial_value, end_value, num_steps

    def get(self, step):
        alpha = step / self.num_steps
        return (1.0 - alpha) * self.initial_value + alpha * self.end_value


class ExponentialScheduler:
    def __init__(self, initial_value, final_value, num_steps):
        self.initial_value, self.final_value, self.num_steps = initial_value, final_value, num_steps
        self.mean_lifetime = -(num_steps - 1) / math.log(final_value / initial_value)

    def get(self, step):
        if step >= self.num_steps:
            return self.final_value
        return self.initial_value * math.exp(-step / self.mean_lifetime)
