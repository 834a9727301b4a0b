"""``DataBlock``: the unit ``encode_block`` consumes and ``decode_block`` returns
(reference scl/core/data_block.py:5-106)."""
from __future__ import annotations

import collections
from typing import List, Set

from .prob_dist import ProbabilityDist

__all__ = ["DataBlock"]


class DataBlock:
    """Thin wrapper over a list of symbols (any hashable)."""

    def __init__(self, data_list: List):
        self.data_list = data_list

    @property
    def size(self) -> int:
        return len(self.data_list)

    def get_alphabet(self) -> Set:
        return set(self.data_list)

    def get_counts(self, order=0) -> dict:
        if order != 0:
            raise NotImplementedError("[order != 0] counts not implemented")
        return dict(collections.Counter(self.data_list))

    def get_empirical_distribution(self, order=0) -> ProbabilityDist:
        if order != 0:
            raise NotImplementedError("[order != 0] empirical counts not implemented")
        n = self.size
        return ProbabilityDist({s: c / n for s, c in self.get_counts().items()})

    def get_entropy(self, order=0):
        if order != 0:
            raise NotImplementedError("[order != 0] Entropy computation not implemented")
        return self.get_empirical_distribution().entropy
