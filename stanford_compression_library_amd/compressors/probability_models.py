"""Frequency models for the arithmetic coder (reference scl/compressors/probability_models.py:15-160).

Same constructor signatures, attributes (``freqs_current``, ``freqs_kplus1_tuple``, ``past_k``) and
``update_model`` as the reference.  As in the reference the model OBJECT is the state of the coder that owns
it: ``ArithmeticEncoder.encode_block`` / ``ArithmeticDecoder.decode_block`` read the object's current counts
(and order-k context) before a block, run the block on the device from exactly that state, and write the
advanced state back -- so a coder used for several blocks, or a model inspected between blocks, behaves like
the reference's (which never resets ``freq_model``, arithmetic_coding.py:52-56,118).  The batched C-ABI entry
points (``scl_aec_*_batch``) instead start every chunk from a fresh copy: ``device_spec`` describes that copy.
"""
from __future__ import annotations

import abc
import copy
from typing import List

import numpy as np

from ..core.prob_dist import Frequencies

__all__ = ["FreqModelBase", "FixedFreqModel", "AdaptiveIIDFreqModel", "AdaptiveOrderKFreqModel"]

KIND_FIXED, KIND_IID, KIND_ORDERK = 0, 1, 2


class FreqModelBase(abc.ABC):
    """freqs_initial: starting frequencies; max_allowed_total_freq: cap the adaptive models respect."""

    _kind = None

    def __init__(self, freqs_initial: Frequencies, max_allowed_total_freq):
        self.freqs_current = copy.deepcopy(freqs_initial)
        self.max_allowed_total_freq = max_allowed_total_freq
        # what a *fresh* copy of this model looks like: the device starts every chunk from it
        self._initial_freq_list = [int(f) for f in freqs_initial.freq_list]
        self._alphabet = freqs_initial.alphabet

    @abc.abstractmethod
    def update_model(self, s):
        raise NotImplementedError

    # -- description handed to the device (no reference counterpart) ---------------------------------
    def device_spec(self) -> dict:
        return dict(kind=self._kind, K=len(self._alphabet), k=0, freq_init=self._initial_freq_list,
                    max_total=int(self.max_allowed_total_freq), alphabet=self._alphabet)

    def export_state(self):
        """(counts uint32 [K], past_k uint32 [0]) -- the canonical host form of include/scl_hip.h"""
        return (np.asarray([int(f) for f in self.freqs_current.freq_list], dtype=np.uint32),
                np.zeros(1, dtype=np.uint32))

    def import_state(self, counts, past_k):
        fd = self.freqs_current.freq_dict
        for sym, f in zip(list(fd.keys()), counts.tolist()):
            fd[sym] = int(f)


class FixedFreqModel(FreqModelBase):
    _kind = KIND_FIXED

    def update_model(self, s):
        """the model never changes (probability_models.py:57-67)"""


class AdaptiveIIDFreqModel(FreqModelBase):
    _kind = KIND_IID

    def update_model(self, s):
        """count the symbol; halve everything (floor 1) once the total reaches the cap (:70-92)"""
        fd = self.freqs_current.freq_dict
        fd[s] += 1
        if self.freqs_current.total_freq >= self.max_allowed_total_freq:
            for sym, f in fd.items():
                fd[sym] = max(f // 2, 1)


class AdaptiveOrderKFreqModel(FreqModelBase):
    """k-th order adaptive model: counts of (k+1)-tuples, all ones initially, context = last k symbol
    indices starting from all zeros (probability_models.py:95-160)."""

    _kind = KIND_ORDERK

    def __init__(self, alphabet: List, k: int, max_allowed_total_freq: int):
        assert k >= 0
        self.k = k
        self.alphabet = alphabet
        self.alphabet_to_idx = {a: i for i, a in enumerate(alphabet)}
        self.freqs_kplus1_tuple = np.ones([len(alphabet)] * (k + 1), dtype=int)
        self.max_allowed_total_freq = max_allowed_total_freq
        self.past_k = [0] * k
        self._alphabet = list(alphabet)

    @property
    def freqs_current(self):
        row = self.freqs_kplus1_tuple[tuple(self.past_k)] if self.k > 0 else self.freqs_kplus1_tuple
        return Frequencies(dict(zip(self.alphabet, np.ravel(row).tolist())))

    def update_model(self, s):
        idx = self.alphabet_to_idx[s]
        cell = (*self.past_k, idx)
        self.freqs_kplus1_tuple[cell] += 1
        if self.k > 0:
            self.past_k = self.past_k[1:] + [idx]
        # the reference's rescale branch tests a single cell against the cap and would raise
        # numpy.AxisError (np.max(scalar, 1)); it is unreachable for blocks below 2**30 symbols
        if self.freqs_kplus1_tuple[cell] >= self.max_allowed_total_freq:
            raise ValueError("order-k rescale is undefined in the reference (np.max(scalar, 1))")

    def device_spec(self) -> dict:
        return dict(kind=self._kind, K=len(self._alphabet), k=int(self.k), freq_init=None,
                    max_total=int(self.max_allowed_total_freq), alphabet=self._alphabet)

    def export_state(self):
        past = np.zeros(max(self.k, 1), dtype=np.uint32)
        past[: self.k] = self.past_k
        return np.ascontiguousarray(self.freqs_kplus1_tuple, dtype=np.uint32).ravel(), past

    def import_state(self, counts, past_k):
        self.freqs_kplus1_tuple[...] = counts.astype(self.freqs_kplus1_tuple.dtype).reshape(
            self.freqs_kplus1_tuple.shape)
        self.past_k = [int(v) for v in past_k[: self.k]]
