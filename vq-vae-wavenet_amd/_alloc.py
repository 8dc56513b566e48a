"""Device-buffer allocation for the workspaces, with a debug switch that makes read-before-write visible.

`VQW_POISON=1`: every buffer obtained through `empty()` is filled with NaN when it is allocated, and the buffers recorded for
a workspace are filled again at the start of every step (`repoison`).  torch's caching allocator hands a fresh
`torch.empty` the previous owner's bytes, so a kernel that reads an element nobody wrote this step (a split-K output
assumed clean, a tail tile, a missing `.zero_()`) otherwise sees plausible stale numbers of an earlier step or test;
poisoned, it produces a NaN deterministically.  Integer buffers (labels, code indices) are left alone: they are used as
addresses, and a poisoned index would turn a read-before-write into an out-of-bounds access instead of a NaN.
"""
import os

import torch

POISON = os.environ.get('VQW_POISON', '0') == '1'
_recording = None


def poison(t):
    if t.is_floating_point():
        t.fill_(float('nan'))


def empty(*shape, **kw):
    t = torch.empty(*shape, **kw)
    if POISON:
        poison(t)
        if _recording is not None:
            _recording.append(t)
    return t


class record:
    """`with record(lst):` appends every buffer `empty()` hands out inside the block to `lst`."""

    def __init__(self, lst):
        self.lst = lst

    def __enter__(self):
        global _recording
        self.prev, _recording = _recording, self.lst
        return self.lst

    def __exit__(self, *exc):
        global _recording
        _recording = self.prev


def repoison(tensors):
    if POISON:
        for t in tensors:
            poison(t)
