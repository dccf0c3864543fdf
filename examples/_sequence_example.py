"""Shared expectations of the sequence examples (``simple_sequence`` and the
dyna variants): the first response of a sequence is input+1, later ones echo
the input; dyna models add the correlation id on the last response and the
string dyna model accumulates."""
from _common import fail

VALUES = [11, 7, 5, 3, 2, 0, 1]


def check(int0, int1, str0, id0, id1, sid, dyna):
    for i in range(len(int0)):
        e0 = 1 if i == 0 else VALUES[i - 1]
        e1 = 101 if i == 0 else -VALUES[i - 1]
        if dyna:
            es = 20 if i == 0 else -VALUES[i - 1] + str0[i - 1]
        else:
            es = 21 if i == 0 else -VALUES[i - 1]
        if dyna and i == len(VALUES):
            e0 += id0
            e1 += id1
            es += int(sid)
        print(f"[{i}] {int0[i]} : {int1[i]} : {str0[i]}")
        if (e0, e1, es) != (int0[i], int1[i], str0[i]):
            fail(f"[ expected ] {e0} : {e1} : {es}")
