"""Small helpers of the hot path (reference scl/utils/misc_utils.py:6-11)."""
import functools

cache = functools.lru_cache(maxsize=None)


def is_power_of_two(x) -> bool:
    """True for 1, 2, 4, ... (integers only; the reference takes log2 in float)."""
    x = int(x)
    return x > 0 and (x & (x - 1)) == 0
