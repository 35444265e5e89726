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


# ---- argparse / misc helpers (reference utils.py:17-31,53-78,107-122) ----------------------------------
def seed_all(seed):
    import random

    import numpy
    import torch
    torch.random.manual_seed(seed)
    numpy.random.seed(seed)
    random.seed(seed)


def list_arg(cast_type=str, delimiter=','):
    def f(s):
        return [cast_type(item) for item in s.split(delimiter)] if len(s) > 0 else []
    return f


def block_config_arg(delimiter=',', group_delimiter=':'):
    from functools import partial
    return partial(parse_block_config, delimiter=delimiter, group_delimiter=group_delimiter)


def list_choices_arg(valid_choices=None):
    def fn(s):
        choices = [str(item) for item in s.split(',')]
        for value in choices:
            if valid_choices is not None and value not in valid_choices:
                raise ValueError(f'Invalid choice {value!s}')
        return choices
    return fn


def flatten_list(l):
    import itertools
    return list(itertools.chain.from_iterable(l))


def _visible_devices():
    import os
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):        # ROCm honours both
        if var in os.environ:
            return os.environ[var].split(',')
    return None


def relative_device_id(abs_device_id):
    ids = _visible_devices()
    if ids is None:
        return abs_device_id
    ids = [int(i) for i in ids]
    if abs_device_id not in ids:
        raise ValueError(f'Device {abs_device_id} is not in the visible-devices list.')
    return ids.index(abs_device_id)


def absolute_device_id(rel_device_id):
    ids = _visible_devices()
    return int(ids[rel_device_id]) if ids is not None else int(rel_device_id)


def pbar(*args, **kwargs):
    import tqdm.auto
    kwargs.setdefault('dynamic_ncols', True)
    return tqdm.auto.tqdm(*args, **kwargs)


def trange(*args, **kwargs):
    import tqdm.auto
    kwargs.setdefault('dynamic_ncols', True)
    return tqdm.auto.trange(*args, **kwargs)


import json as _json


class MyEncoder(_json.JSONEncoder):
    """JSON encoder that writes paths as strings and tensors as nested lists (reference utils.py:97-104)."""

    def default(self, obj):
        import pathlib

        import torch
        if isinstance(obj, pathlib.PurePath):
            return str(obj)
        if torch.is_tensor(obj):
            return obj.tolist()
        return _json.JSONEncoder.default(self, obj)
