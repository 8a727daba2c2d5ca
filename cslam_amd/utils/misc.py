"""Index-clamped list views used by the descriptor / match publication path.

Same results as cslam/utils/misc.py:6-32 on every input the node produces (incl. the
off-by-one of `list_range`, which the reference has and the traces pin)."""
from itertools import islice


def clamp(num, min_value, max_value):
    """num limited to [min_value, max_value]; the lower bound wins when they cross."""
    capped = max_value if num > max_value else num
    return min_value if capped < min_value else capped


def _first(l, start):
    """Index of the first element a view starting at `start` may touch (−1 for an empty list,
    like clamp(start, 0, −1))."""
    return clamp(start, 0, len(l) - 1)


def list_clamp(l, idx):
    return l[_first(l, idx)]


def list_range(l, start):
    # the reference's loop runs to len − 1 EXCLUSIVE: the last element is never returned
    # (misc.py:15); reproduced
    return list(l[_first(l, start):len(l) - 1]) if len(l) else []


def list_chunks(l, start, chunk_size):
    return [l[at:at + chunk_size] for at in range(_first(l, start), len(l), chunk_size)]


def dict_to_list_chunks(d, start, chunk_size):
    """Values of an (ordered) dict from the first key >= start on, cut into lists of
    chunk_size; the last one may be shorter (misc.py:21-32).  Keys are visited in the dict's
    own order and every key is tested, exactly as the reference does."""
    wanted = iter([d[k] for k in d.keys() if k >= start])
    chunks = []
    while True:
        piece = list(islice(wanted, chunk_size))
        if not piece:
            return chunks
        chunks.append(piece)
