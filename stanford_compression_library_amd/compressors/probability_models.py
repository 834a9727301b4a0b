"""Frequency models for the arithmetic coder (reference scl/compressors/probability_models.py:15-160).

On the device each chunk owns a private copy of the model state (``csrc/scl_aec.hip``: ``LaneModel``);
these classes are the host-side description of *which* model to instantiate per chunk, with the
reference's constructor signatures.  They also keep a small host mirror (``freqs_current`` /
``update_model``) so code that inspects a model between symbols keeps working; the coders never use the
mirror to produce bits.
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
