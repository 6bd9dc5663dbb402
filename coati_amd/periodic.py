"""(group, period) layout of the periodic table as the reference's XY_ONE_HOT_FULL uses it
(coati/common/periodic_table.py:3907-3921): one-hot width 28 = 18 groups + 10 rows, lanthanides on row 9,
actinides on row 10, index 0 = pad with (-1,-1) (python negative indexing -> hot indices 27 and 17)."""
_NOBLE = (2, 10, 18, 36, 54, 86, 118)
ONE_HOT_WIDTH = 28


def xy_position(z: int):
    if z == 0:
        return -1, -1
    if 57 <= z <= 71:
        return z - 54, 9
    if 89 <= z <= 103:
        return z - 86, 10
    start, period = 1, 8
    for p, last in enumerate(_NOBLE, start=1):
        if z <= last:
            period = p
            break
        start = last + 1
    else:
        start = 119
    k = z - start
    if period == 1:
        x = 1 if k == 0 else 18
    elif period in (2, 3):
        x = k + 1 if k < 2 else k + 11
    elif period in (4, 5, 8):
        x = k + 1
    else:  # 6, 7: the f-block elements are handled above
        x = k + 1 if k < 2 else k - 13
    return x, period


def onehot_lut(n=120):
    """Two int lists (ix, iy): hot indices of element z; -1 where the reference's 28-wide vector would overflow
    (ypos = 10, the actinides: the reference raises IndexError there)."""
    ix, iy = [], []
    for z in range(n):
        x, y = xy_position(z)
        ix.append(x % ONE_HOT_WIDTH)
        j = 18 + y
        iy.append(j % ONE_HOT_WIDTH if j < ONE_HOT_WIDTH else -1)
    return ix, iy
