#!/usr/bin/env python3
"""Generate (and check) the tables of board_ops.hpp's plane-space transition (transition_oriented):
ORIENT_SELECTORS[o][a] -- selectors of ONE pass of the two-stage v_perm_b32 network that takes a board held as the planes of
action o to the planes of action a -- and ORIENT_RANK_MASKS[a] -- the "first half" cell masks of the row-major rank walk in the
bit layout of the planes of action a.  Everything is derived from DIR_SELECTORS (the forward / inverse networks the step kernel
uses) by pushing labelled cells through the network; each solved selector set is verified by applying it.
    python tools/gen_orientation_tables.py        # prints the two tables as they stand in board_ops.hpp
"""
import itertools, re
vals = """0x06020400u, 0x07030501u, 0x05040100u, 0x07060302u, 0x06020400u, 0x07030501u, 0x05040100u, 0x07060302u,
0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u,
0x05010703u, 0x04000602u, 0x05040100u, 0x07060302u, 0x02060004u, 0x03070105u, 0x01000504u, 0x03020706u,
0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u, 0x07060504u, 0x03020100u"""
D = [int(x.strip().rstrip('u'), 16) for x in vals.replace('\n', ' ').split(',')]
def perm(hi, lo, sel):  # hi, lo: lists of 4 symbols; sel: int
    src = lo + hi
    return [src[(sel >> (8 * i)) & 0xFF] for i in range(4)]
def network(x, s):
    sA, sB, sC, sD = s
    t0, t2 = perm(x[1], x[0], sA), perm(x[3], x[2], sA)
    t1, t3 = perm(x[1], x[0], sB), perm(x[3], x[2], sB)
    return [perm(t2, t0, sC), perm(t3, t1, sC), perm(t2, t0, sD), perm(t3, t1, sD)]
board = [[(r, c) for c in range(4)] for r in range(4)]
planes = {}
for a in range(4):
    planes[a] = network(board, D[8 * a:8 * a + 4])
    back = network(planes[a], D[8 * a + 4:8 * a + 8])
    assert back == board, a
    print('action', a, planes[a])
def solve(target, src):
    """selectors so that network(src) == target (both 4x4 of symbols)"""
    pos = {src[k][b]: (k, b) for k in range(4) for b in range(4)}
    # requests: for byte b and stage-2 selector S in (C, D): (h, codeA, codeB)
    req = {}
    for S, (ya, yb) in (('C', (0, 1)), ('D', (2, 3))):
        for b in range(4):
            ka, ja = pos[target[ya][b]]
            kb, jb = pos[target[yb][b]]
            if (ka >= 2) != (kb >= 2):
                return None
            req[(S, b)] = (ka >= 2, 4 * (ka & 1) + ja, 4 * (kb & 1) + jb)
    pairs = sorted({(r[1], r[2]) for r in req.values()})
    if len(pairs) > 4:
        return None
    slot = {p: i for i, p in enumerate(pairs)}
    sA = [0, 1, 2, 3]; sB = [0, 1, 2, 3]
    for p, i in slot.items():
        sA[i], sB[i] = p
    sel = {}
    for S in 'CD':
        sel[S] = [4 * req[(S, b)][0] + slot[(req[(S, b)][1], req[(S, b)][2])] for b in range(4)]
    pack = lambda l: sum(v << (8 * i) for i, v in enumerate(l))
    s = (pack(sA), pack(sB), pack(sel['C']), pack(sel['D']))
    assert network(src, s) == target
    return s
comp = {}
for o in range(4):
    for a in range(4):
        s = solve(planes[a], planes[o])
        comp[(o, a)] = s
        print(o, a, None if s is None else [hex(v) for v in s])
ROWBIT = [5, 6, 7, 4]
masks = {}
for a in range(4):
    M = [0, 0, 0, 0]
    for k in range(4):
        for b in range(4):
            r, c = planes[a][k][b]
            bit = 1 << (8 * b + ROWBIT[k])
            if r in (0, 1): M[0] |= bit
            if r in (0, 2): M[1] |= bit
            if c in (0, 1): M[2] |= bit
            if c in (0, 2): M[3] |= bit
    masks[a] = M
    print('masks', a, [hex(m) for m in M])
print('\nconstexpr uint32_t ORIENT_SELECTORS[64] = {')
for o in range(4):
    for a in range(4):
        print('    /* %d -> %d */ ' % (o, a) + ', '.join('0x%08Xu' % v for v in comp[(o, a)]) + ',')
print('};\nconstexpr uint32_t ORIENT_RANK_MASKS[16] = {')
for a in range(4):
    print('    /* %d */ ' % a + ', '.join('0x%08Xu' % v for v in masks[a]) + ',')
print('};')
