""" Point samplers with the `batchflow.sampler` surface pydens re-exports (reference pydens/__init__.py:5):
`NumpySampler(name, **kwargs)`, `a & b` = column concatenation, `.sample(size) -> ndarray [size, dim]`
(call site reference model_torch.py:433; README.md:82 `NumpySampler('uniform') & NumpySampler('uniform', low=1, high=5)`).

`batchflow` itself is third-party and not vendored by the reference; only the numpy-distribution samplers and
`&` are restated. On top of the numpy surface every sampler offers `sample_device(size, device)`, which draws the
batch directly in HBM (uniform / normal), so that `Solver.fit` keeps the host out of the step path, and `columns()`:
the per-column description (kind, a, b) of include/pinn.h `pinn_sample_points` when the sampler is a product of
independent uniform / normal / constant columns -- `Solver.fit` then draws the whole batch with ONE launch of the
Philox kernel instead of a handful of torch ops per iteration.
"""
import numpy as np
import torch

__all__ = ['Sampler', 'NumpySampler', 'ConstantSampler', 'NS']

_ALIASES = {'u': 'uniform', 'n': 'normal', 'e': 'exponential', 'g': 'gamma'}


class Sampler:
    dim = 1

    def sample(self, size):
        raise NotImplementedError

    def sample_device(self, size, device, generator=None):
        """ default: numpy draw + one host-to-device copy. """
        return torch.from_numpy(np.asarray(self.sample(size), dtype=np.float32)).to(device)

    def columns(self):
        """ [(kind, a, b)] per column for the device sampler kernel (0 uniform on [a, b), 1 normal a + b z, 2 constant a),
        or None when the sampler is not such a product (then `sample_device` / `sample` are used). """
        return None

    def seeds(self):
        """ seeds of the random leaves of this sampler, None for an unseeded leaf (constants contribute nothing) """
        return []

    def device_key(self):
        """ Philox key for the device sampler when EVERY random leaf was given a seed (`NumpySampler(..., seed=k)`: the
        reference then draws a sequence that depends on those seeds alone, model_torch.py:433), else None: the solver
        keys the stream by torch's generator instead. """
        seeds = self.seeds()
        if not seeds or any(s is None for s in seeds):
            return None
        key = 0x9E3779B97F4A7C15
        try:
            for s in seeds:                       # splitmix64-style fold
                key = (key ^ (int(s) & (2 ** 64 - 1))) * 0xBF58476D1CE4E5B9 & (2 ** 64 - 1)
                key ^= key >> 31
        except (TypeError, ValueError):           # array-like seeds (numpy accepts them): no single integer to key a stream with
            return None
        return key

    def next_device_call(self, count=1):
        """ batches this sampler object has drawn on the device so far (its own call counter, like its own numpy state);
        reserves `count` consecutive batch numbers and returns the first """
        n = getattr(self, '_device_calls', 0)
        self._device_calls = n + count
        return n

    def __and__(self, other):
        if isinstance(other, (int, float)):
            other = ConstantSampler(other)
        return _ConcatSampler(self, other)


class _ConcatSampler(Sampler):
    def __init__(self, left, right):
        self.left, self.right = left, right
        self.dim = left.dim + right.dim

    def sample(self, size):
        return np.concatenate([self.left.sample(size), self.right.sample(size)], axis=1)

    def sample_device(self, size, device, generator=None):
        return torch.cat([self.left.sample_device(size, device, generator),
                          self.right.sample_device(size, device, generator)], dim=1)

    def columns(self):
        left, right = self.left.columns(), self.right.columns()
        return None if left is None or right is None else left + right

    def seeds(self):
        return self.left.seeds() + self.right.seeds()


class ConstantSampler(Sampler):
    def __init__(self, value, dim=1):
        self.value, self.dim = float(value), dim

    def sample(self, size):
        return np.full((size, self.dim), self.value, dtype=np.float64)

    def sample_device(self, size, device, generator=None):
        return torch.full((size, self.dim), self.value, dtype=torch.float32, device=device)

    def columns(self):
        return [(2, self.value, 0.0)] * self.dim


class NumpySampler(Sampler):
    """ sampler named after a `numpy.random` distribution ('uniform'/'u', 'normal'/'n', ...).

    `seed=`: `sample()` (host numpy) replays numpy's own stream, as batchflow's sampler does. Inside `Solver.fit` products of
    uniform / normal / constant columns are drawn ON THE DEVICE by a counter-based Philox generator keyed with the seeds
    (`device_key`): runs are reproducible for a given seed, rank and batch count, but the points are NOT numpy's -- two
    leaves with the same seed give independent columns there, every rank draws its own points -- so a seeded run does not
    reproduce the reference's point sets bit for bit. For that, hand `fit` a sampler object without `columns()` (e.g. a
    thin wrapper exposing only `sample`): it then takes the host path of the reference (model_torch.py:433). """
    def __init__(self, name, dim=1, seed=None, **kwargs):
        self.name = _ALIASES.get(name, name)
        self.dim = dim
        self.kwargs = kwargs
        self.seed = seed
        self.rng = np.random.RandomState(seed)
        if not hasattr(self.rng, self.name):
            raise ValueError(f'unknown numpy distribution {name!r}')

    def seeds(self):
        return [self.seed] * self.dim

    def sample(self, size):
        draw = getattr(self.rng, self.name)
        return np.asarray(draw(size=(size, self.dim), **self.kwargs), dtype=np.float64)

    def sample_device(self, size, device, generator=None):
        if self.name == 'uniform' and not (set(self.kwargs) - {'low', 'high'}):
            low, high = float(self.kwargs.get('low', 0.0)), float(self.kwargs.get('high', 1.0))
            out = torch.rand((size, self.dim), dtype=torch.float32, device=device, generator=generator)
            if low != 0.0 or high != 1.0:
                out = out * (high - low) + low
            return out
        if self.name == 'normal' and not (set(self.kwargs) - {'loc', 'scale'}):
            out = torch.randn((size, self.dim), dtype=torch.float32, device=device, generator=generator)
            return out * float(self.kwargs.get('scale', 1.0)) + float(self.kwargs.get('loc', 0.0))
        return super().sample_device(size, device, generator)

    def columns(self):
        if self.name == 'uniform' and not (set(self.kwargs) - {'low', 'high'}):
            low, high = self.kwargs.get('low', 0.0), self.kwargs.get('high', 1.0)
            if np.ndim(low) == 0 and np.ndim(high) == 0:
                return [(0, float(low), float(high))] * self.dim
        if self.name == 'normal' and not (set(self.kwargs) - {'loc', 'scale'}):
            loc, scale = self.kwargs.get('loc', 0.0), self.kwargs.get('scale', 1.0)
            if np.ndim(loc) == 0 and np.ndim(scale) == 0:
                return [(1, float(loc), float(scale))] * self.dim
        return None


NS = NumpySampler
