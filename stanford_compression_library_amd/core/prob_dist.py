"""Model tables of the hot path: ``Frequencies`` (+ the ``ProbabilityDist`` the tests sample from).

Mirrors the public surface of reference scl/core/prob_dist.py (``ProbabilityDist`` :6-90,
``get_avg_neg_log_prob`` :143-158, ``Frequencies`` :161-230).  Symbol order is the *insertion
order* of the dict (quirk Q8); the device sees symbols only as indices into ``alphabet``.
"""
from __future__ import annotations

import math
from typing import Dict, Hashable, List

import numpy as np

__all__ = ["ProbabilityDist", "Frequencies", "get_avg_neg_log_prob"]


class ProbabilityDist:
    """Ordered symbol -> probability map (reference scl/core/prob_dist.py:6-90)."""

    def __init__(self, prob_dict: Dict[Hashable, float] = None):
        self._validate_prob_dist(prob_dict)
        self.prob_dict = prob_dict

    @staticmethod
    def _validate_prob_dist(prob_dict) -> None:
        """every probability at least 1e-6 (AssertionError) and the sum within 1e-8 of one (ValueError) -- the checks and
        the exception types of reference scl/core/prob_dist.py:77-90"""
        total = 0.0
        for p in prob_dict.values():
            assert p >= 1e-6, "probabilities negative or too small cause stability issues"
            total += p
        if abs(total - 1.0) > 1e-8:
            raise ValueError("probabilities do not sum to 1")

    def __repr__(self):
        return f"ProbabilityDist({self.prob_dict!r})"

    @property
    def size(self) -> int:
        return len(self.prob_dict)

    @property
    def alphabet(self) -> List:
        return list(self.prob_dict)

    @property
    def prob_list(self) -> List[float]:
        return list(self.prob_dict.values())

    @classmethod
    def get_sorted_prob_dist(cls, prob_dict, descending=False):
        return cls(dict(sorted(prob_dict.items(), key=lambda kv: kv[1], reverse=descending)))

    @classmethod
    def normalize_prob_dict(cls, prob_dict):
        z = sum(prob_dict.values())
        return cls({s: p / z for s, p in prob_dict.items()})

    @property
    def cumulative_prob_dict(self) -> Dict[Hashable, float]:
        out, acc = {}, 0
        for s, p in self.prob_dict.items():
            out[s] = acc
            acc += p
        return out

    @property
    def entropy(self) -> float:
        h = 0
        for p in self.prob_dict.values():
            h += -p * np.log2(p)
        return h

    def probability(self, symbol) -> float:
        return self.prob_dict[symbol]

    def neg_log_probability(self, symbol) -> float:
        return -np.log2(self.prob_dict[symbol])


def get_avg_neg_log_prob(prob_dist: ProbabilityDist, data_block) -> float:
    """Average -log2 p(symbol) over a block (reference scl/core/prob_dist.py:143-158)."""
    nlp = {s: -math.log2(p) for s, p in prob_dist.prob_dict.items()}
    return sum(nlp[s] for s in data_block.data_list) / data_block.size


class Frequencies:
    """Ordered symbol -> integer count map (reference scl/core/prob_dist.py:161-230).

    The reference recomputes ``cumulative_freq_dict`` on every access (28-34 % of its coder time,
    SURVEY.md 3.1); here the derived views are cheap one-shot computations and the device tables
    come from :meth:`index_tables`.
    """

    def __init__(self, freq_dict: Dict[Hashable, int] = None):
        self.freq_dict = freq_dict

    def __repr__(self):
        return f"Frequencies({self.freq_dict!r})"

    @property
    def size(self) -> int:
        return len(self.freq_dict)

    @property
    def alphabet(self) -> List:
        return list(self.freq_dict)

    @property
    def freq_list(self) -> List[int]:
        return list(self.freq_dict.values())

    @property
    def total_freq(self) -> int:
        # the reference returns numpy.int64 (np.sum); a Python int is value-identical for every
        # parameter set accepted here (H < 2**63 is asserted by the params dataclasses, quirk Q7)
        return int(sum(int(f) for f in self.freq_dict.values()))

    @property
    def cumulative_freq_dict(self) -> Dict[Hashable, int]:
        out, acc = {}, 0
        for s, f in self.freq_dict.items():
            out[s] = acc
            acc += f
        return out

    def frequency(self, symbol) -> int:
        return self.freq_dict[symbol]

    def get_prob_dist(self) -> ProbabilityDist:
        m = self.total_freq
        return ProbabilityDist({s: f / m for s, f in self.freq_dict.items()})

    # ---- device-facing views (no reference counterpart) ---------------------------------------
    def index_tables(self):
        """(freq[K], cum[K]) as uint32 arrays in alphabet (insertion) order."""
        f = np.asarray([int(v) for v in self.freq_dict.values()], dtype=np.int64)
        c = np.concatenate([[0], np.cumsum(f)[:-1]]) if f.size else f
        return f.astype(np.uint32), c.astype(np.uint32)

    def symbol_index(self) -> Dict[Hashable, int]:
        return {s: i for i, s in enumerate(self.freq_dict)}

    @staticmethod
    def _validate_freq_dist(freq_dict):
        for f in freq_dict.values():
            assert f > 0, "frequency cannot be negative or 0"
            assert isinstance(f, (int, np.integer))
