""" Stand-in for `batchflow.models.torch.Block` (call site: reference pydens/model_torch.py:164-168).

Documented behaviour restated (reference model_torch.py:142-156): the layout string is a sequence of letters,
'f' = fully connected layer taking its width from `features`, 'a' = activation (one shared name / class, or a
sequence with one entry per 'a'), 'R' = start of a skip connection (remembers the current tensor), '+' = end of it
(adds the remembered tensor); spaces are ignored. 'Sin' is accepted as an activation name (torch.nn has no such
module; the docstring at :153 lists it). The real batchflow is absent from this image -- "parity unpinned" at this
boundary -- so these semantics come from that docstring alone.
"""
import torch
from torch import nn


class Sin(nn.Module):
    def forward(self, x):
        return torch.sin(x)


class Lambda(nn.Module):
    """ a plain callable as an activation (the docstring at :150 says "Sequence of callables, str") """
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


def make_activation(act):
    if isinstance(act, str):
        act = Sin if act == 'Sin' else getattr(nn, act)
    if isinstance(act, type):
        return act()
    if isinstance(act, nn.Module):
        return act
    if act is torch.sin:
        return Sin()
    if callable(act):
        return Lambda(act)
    raise NotImplementedError(f'activation {act!r}')


class Block(nn.Module):
    """ iterable over its layers like nn.Sequential; `forward` follows the layout letters. """
    def __init__(self, inputs=None, layout='', features=(), activation='Sigmoid', **kwargs):
        _ = kwargs
        super().__init__()
        n_in = inputs.shape[1]
        features = list(features)
        acts = list(activation) if isinstance(activation, (list, tuple)) else None
        self.layers = nn.ModuleList()
        self.program = []                       # one entry per letter: index into self.layers, 'R' or '+'
        for letter in layout.replace(' ', ''):
            if letter == 'f':
                n_out = features.pop(0)
                self.layers.append(nn.Linear(n_in, n_out, bias=True))
                n_in = n_out
            elif letter == 'a':
                self.layers.append(make_activation(acts.pop(0) if acts is not None else activation))
            elif letter not in 'R+':
                raise NotImplementedError(f"layout letter {letter!r} is outside the stand-in's scope ('f', 'a', 'R', '+')")
            self.program.append(letter if letter in 'R+' else len(self.layers) - 1)

    def __iter__(self):
        return iter(self.layers)

    def __len__(self):
        return len(self.layers)

    def forward(self, x):
        stack = []
        for step in self.program:
            if step == 'R':
                stack.append(x)
            elif step == '+':
                x = x + stack.pop()
            else:
                x = self.layers[step](x)
        return x
