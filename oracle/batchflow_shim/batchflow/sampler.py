""" Stand-in for `batchflow.sampler` (re-exported by reference pydens/__init__.py:5; used at
model_torch.py:433 as `sampler.sample(batch_size) -> ndarray [size, total]`, README.md:82 as
`NumpySampler('uniform') & NumpySampler('uniform', low=1, high=5)`).

Restated: numpy-RNG samplers named after `numpy.random` distributions ('u' is the documented alias
of 'uniform', tutorial cell `NS('u') & NS('u', low=.5, high=5.5)`), `&` concatenates columns,
`dim=k` draws k i.i.d. columns.
"""
import numpy as np

__all__ = ['Sampler', 'NumpySampler', 'NS']

_ALIASES = {'u': 'uniform', 'n': 'normal', 'e': 'exponential', 'g': 'gamma'}


class Sampler:
    dim = 1

    def sample(self, size):
        raise NotImplementedError

    def __and__(self, other):
        return _ConcatSampler(self, other)


class _ConcatSampler(Sampler):
    def __init__(self, left, right):
        self.left, self.right = left, right
        self.dim = left.dim + right.dim

    def sample(self, size):
        return np.concatenate([self.left.sample(size), self.right.sample(size)], axis=1)


class NumpySampler(Sampler):
    def __init__(self, name, dim=1, seed=None, **kwargs):
        self.name = _ALIASES.get(name, name)
        self.dim = dim
        self.kwargs = kwargs
        self.rng = np.random.RandomState(seed)

    def sample(self, size):
        draw = getattr(self.rng, self.name)
        return np.asarray(draw(size=(size, self.dim), **self.kwargs), dtype=np.float64)


NS = NumpySampler
