"""Minimal hyper-parameter store: the subset of `varz.Vars` the reference relies on.

Used by the reference at /root/reference/gpar/regression.py:101-173 (`vs.bnd`, `vs.get`), :335-336
(`vs.names`, `vs[name]`), :454-459 (glob names handed to the optimiser) and by its tests
(`vs.copy(detach=True)`, `vs.requires_grad(True)`, `vs.get_vars()`; tests/test_regression.py:88,153-158,236).

A variable is stored as an unconstrained ("latent") float64 torch tensor on the CPU.  `get` returns the latent
itself, `pos` its exponential, `bnd` the map  lower + (upper - lower) * sigmoid(latent)  (varz's bounded
transform; default bounds [1e-4, 1e4]).  The constrained value is a differentiable function of the latent, so
gradients of anything computed from it flow back to `.grad` of the latent.  Get-or-create semantics: once a
name exists, `init` and the bounds passed later are ignored (varz behaviour, relied upon by scale tying:
regression.py:102-105).
"""
import fnmatch
from collections import OrderedDict

import numpy as np
import torch

__all__ = ["Vars"]


class _Var:
    __slots__ = ("latent", "kind", "lower", "upper", "_cached", "_cached_version")

    def __init__(self, latent, kind, lower=None, upper=None):
        self.latent, self.kind, self.lower, self.upper = latent, kind, lower, upper
        self._cached, self._cached_version = None, -1

    def value(self):
        if self.kind == "get":
            return self.latent
        # Outside autograd the constrained value only changes when the latent is written (in place: its version counter
        # moves): an evaluation of a p-layer model otherwise spends ~40 tiny host-side torch operators per layer here
        # (0.2 of the 0.36 ms a layer of C2 costs the host).
        track = torch.is_grad_enabled() and self.latent.requires_grad
        if not track and self._cached is not None and self._cached_version == self.latent._version:
            return self._cached
        if self.kind == "pos":
            value = torch.exp(self.latent)
        else:
            value = self.lower + (self.upper - self.lower) * torch.sigmoid(self.latent)
        if not track:
            self._cached, self._cached_version = value.detach(), self.latent._version
            return self._cached
        return value


class Vars:
    def __init__(self, dtype=torch.float64):
        self.dtype = dtype
        self._vars = OrderedDict()
        self._requires_grad = False
        self._memo = {}

    def epoch(self):
        """A number that changes whenever a variable is created or written in place; None while autograd is tracking any of
        them (objects built from the values then belong to one graph and must not be reused)."""
        tracking = torch.is_grad_enabled()
        total = len(self._vars)
        for var in self._vars.values():
            if tracking and var.latent.requires_grad:
                return None
            total += var.latent._version
        return total

    def memo(self, key, build):
        """`build()` once per epoch: layer constructors are pure functions of the stored values, and an evaluation of a
        p-layer model otherwise rebuilds ~15 kernel objects per layer (0.1 ms of host time each layer, which at n = 4096 is what
        delays the last of four pipelined layers)."""
        epoch = self.epoch()
        if epoch is None:
            return build()
        hit = self._memo.get(key)
        if hit is not None and hit[0] == epoch:
            return hit[1]
        value = build()
        after = self.epoch()   # (the first build creates the variables)
        if after is not None:
            self._memo[key] = (after, value)
        return value

    # ---- creation ------------------------------------------------------------------------------
    def _new(self, name, latent_value, kind, lower=None, upper=None):
        latent = torch.tensor(np.asarray(latent_value, dtype=np.float64), dtype=self.dtype)
        latent.requires_grad_(self._requires_grad)
        self._vars[name] = _Var(latent, kind, lower, upper)
        return self._vars[name]

    def get(self, init=None, name=None, shape=None):
        """Unbounded variable."""
        if name is None:
            raise ValueError("variables must be named")
        if name not in self._vars:
            if init is None:
                init = np.random.randn(*(shape or ()))
            self._new(name, init, "get")
        return self._vars[name].value()

    unbounded = get

    def pos(self, init=None, name=None, shape=None):
        """Positive variable (exp transform)."""
        if name is None:
            raise ValueError("variables must be named")
        if name not in self._vars:
            if init is None:
                init = np.random.rand(*(shape or ()))
            self._new(name, np.log(np.asarray(init, dtype=np.float64)), "pos")
        return self._vars[name].value()

    positive = pos

    def bnd(self, init=None, lower=1e-4, upper=1e4, name=None, shape=None):
        """Variable constrained to (lower, upper) through a sigmoid."""
        if name is None:
            raise ValueError("variables must be named")
        if name not in self._vars:
            if init is None:
                init = lower + (upper - lower) * np.random.rand(*(shape or ()))
            init = np.asarray(init, dtype=np.float64)
            if np.any(init < lower) or np.any(init > upper):
                raise ValueError(f'initial value of "{name}" must lie inside [{lower}, {upper}]')
            with np.errstate(divide="ignore"):
                # an initial value ON a bound maps to an infinite latent, as in varz (the reference's tests use
                # noise=1e-8 with lower=1e-8: tests/test_regression.py:147,169)
                latent = np.log(init - lower) - np.log(upper - init)
            self._new(name, latent, "bnd", float(lower), float(upper))
        return self._vars[name].value()

    bounded = bnd

    # ---- access --------------------------------------------------------------------------------
    @property
    def names(self):
        return list(self._vars.keys())

    def __contains__(self, name):
        return name in self._vars

    def __getitem__(self, name):
        return self._vars[name].value()

    def assign(self, name, value):
        """Set the constrained value of an existing variable."""
        var = self._vars[name]
        value = np.asarray(value, dtype=np.float64)
        if var.kind == "get":
            latent = value
        elif var.kind == "pos":
            latent = np.log(value)
        else:
            with np.errstate(divide="ignore"):
                latent = np.log(value - var.lower) - np.log(var.upper - value)
        with torch.no_grad():
            var.latent.copy_(torch.tensor(latent, dtype=self.dtype).reshape(var.latent.shape))

    def match(self, patterns):
        """Names matching any of the glob patterns (e.g. "0/*"), in creation order."""
        if isinstance(patterns, str):
            patterns = [patterns]
        return [n for n in self._vars if any(fnmatch.fnmatchcase(n, p) for p in patterns)]

    def get_vars(self, *names):
        """Latent (unconstrained) tensors, all or those matching the given names / globs."""
        selected = self.match(list(names)) if names else self.names
        return [self._vars[n].latent for n in selected]

    def requires_grad(self, value, *names):
        selected = self.match(list(names)) if names else self.names
        if not names:
            self._requires_grad = bool(value)
        for n in selected:
            self._vars[n].latent.requires_grad_(bool(value))

    def copy(self, detach=False):
        out = Vars(dtype=self.dtype)
        out._requires_grad = False if detach else self._requires_grad
        for name, var in self._vars.items():
            latent = var.latent.detach().clone()
            if not detach:
                latent.requires_grad_(var.latent.requires_grad)
            out._vars[name] = _Var(latent, var.kind, var.lower, var.upper)
        return out

    def detach(self):
        return self.copy(detach=True)

    # ---- flat packing for the optimiser ----------------------------------------------------------
    def get_vector(self, names):
        return np.concatenate([self._vars[n].latent.detach().numpy().reshape(-1) for n in names]) if names else np.zeros(0)

    def set_vector(self, vector, names):
        vector = np.asarray(vector, dtype=np.float64)
        i = 0
        with torch.no_grad():
            for n in names:
                latent = self._vars[n].latent
                size = latent.numel()
                latent.copy_(torch.tensor(vector[i : i + size], dtype=self.dtype).reshape(latent.shape))
                i += size

    def print(self):  # pragma: no cover - convenience
        for n in self.names:
            print(f"{n}: {self[n].detach().numpy()}")
