""" Stand-in for `batchflow.models.torch.Block` (call site: reference pydens/model_torch.py:164-168).

Documented behaviour restated (reference model_torch.py:142-156): the layout string is a sequence of
letters, 'f' = fully connected layer taking its width from `features`, 'a' = activation; spaces are
ignored. Only these two letters are in scope.
"""
import torch
from torch import nn


class Block(nn.Sequential):
    def __init__(self, inputs=None, layout='', features=(), activation='Sigmoid', **kwargs):
        _ = kwargs
        n_in = inputs.shape[1]
        features = list(features)
        layers = []
        acts = list(activation) if isinstance(activation, (list, tuple)) else None
        for letter in layout.replace(' ', ''):
            if letter == 'f':
                n_out = features.pop(0)
                layers.append(nn.Linear(n_in, n_out, bias=True))
                n_in = n_out
            elif letter == 'a':
                act = acts.pop(0) if acts is not None else activation
                if isinstance(act, str):
                    act = getattr(nn, act)
                layers.append(act() if isinstance(act, type) else act)
            else:
                raise NotImplementedError(f"layout letter {letter!r} is outside the stand-in's scope ('f', 'a')")
        super().__init__(*layers)
