"""Flat fp32 parameter arenas behind nn.Module parameters.

Every network keeps its weights in ONE contiguous device buffer laid out by the C library
(hypad_param_info); the nn.Parameters the user sees (reference ``state_dict`` names and shapes,
SURVEY.md A.1) are views into it, so the fused kernels take a single pointer per network.
"""
import torch
from torch import nn

from . import _C


class _Group(nn.Module):
    """Name-space holder so that parameters appear as e.g. ``lstm.weight_ih_l0``."""


class ArenaModule(nn.Module):
    _net = None

    def _init_arena(self, net, S, L, hyperbolic, init):
        self._net, self._dims = net, (int(S), int(L), int(bool(hyperbolic)))
        cat, total = _C.param_catalogue(net, S, L, hyperbolic)
        self._catalogue, self._total = cat, total
        arena = torch.zeros(total, dtype=torch.float32)
        self._slots = {}
        for name, off, shape in cat:
            n = 1
            for s in shape:
                n *= s
            view = arena[off:off + n].view(shape)
            view.copy_(init[name].detach().to(torch.float32).reshape(shape))
            holder, leaf = self._holder_for(name)
            existing = getattr(holder, leaf, None) if leaf in getattr(holder, "_parameters", {}) else None
            if existing is not None:           # keep Parameter subclass / identity (e.g. the ball-valued bias)
                existing.data = view
                param = existing
            else:
                param = nn.Parameter(view)
                holder.register_parameter(leaf, param)
            self._slots[name] = (param, off, n, shape)
        self._arena = arena
        self._arena_calls = 0

    def _holder_for(self, name):
        parts = name.split(".")
        holder = self
        for p in parts[:-1]:
            if not hasattr(holder, p):
                setattr(holder, p, _Group())
            holder = getattr(holder, p)
        return holder, parts[-1]

    # ---- keep the views glued to the arena across .cuda()/.to()/load_state_dict()/manual reassignment
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._repack()
        return self

    def _repack(self):
        first = next(iter(self._slots.values()))[0]
        dev = first.device
        for p, _, _, _ in self._slots.values():
            if p.dtype != torch.float32:
                raise _C.HypadError("hypad_amd networks are fp32 (the reference runs them in fp32, models/tadgan.py:24,92)")
        arena = torch.zeros(self._total, dtype=torch.float32, device=dev)
        for name, (p, off, n, shape) in self._slots.items():
            holder, leaf = self._holder_for(name)
            cur = getattr(holder, leaf)
            view = arena[off:off + n].view(shape)
            view.copy_(cur.detach().to(dev))
            cur.data = view
            self._slots[name] = (cur, off, n, shape)
        self._arena = arena
        self._glue_items = None

    def _glued(self, spot=False):
        base = self._arena.data_ptr()
        dev = self._arena.device
        items = getattr(self, "_glue_items", None)
        if items is None or len(items) != len(self._slots):      # (holder's parameter dict, leaf name) resolved once
            items = self._glue_items = [(self._holder_for(name)[0]._parameters, self._holder_for(name)[1], name) for name in self._slots]
        if spot:                                                  # first and last tensor only
            items = (items[0], items[-1])
        for params, leaf, name in items:
            p, off, n, shape = self._slots[name]
            cur = params.get(leaf)
            if cur is not p or cur.data_ptr() != base + 4 * off or cur.device != dev:
                return False
        return True

    FULL_CHECK_EVERY = 32

    def arena(self, fast=False):
        """The flat device buffer, re-packed first if a parameter was re-assigned behind our back (``module.w = Parameter(..)``
        or ``p.data = ..``; ``copy_`` / ``load_state_dict`` / ``.to()`` keep the views glued by themselves).  Verifying every
        view costs ~1 us per tensor -- more than an iteration's launches for the decoder -- so the per-iteration callers
        (``hypad_amd.train``, ``fast=True``) get the full verification on every 32nd call and a two-tensor spot check
        otherwise; everyone else gets it on every call."""
        self._arena_calls = getattr(self, "_arena_calls", 0) + 1
        full = (not fast) or self._arena_calls % self.FULL_CHECK_EVERY == 1
        if not self._glued(spot=not full):
            self._repack()
        if not self._arena.is_cuda:
            raise _C.HypadError("network parameters must live on the GPU: call .cuda() (hypad_amd has no CPU path)")
        return self._arena
