#!/usr/bin/env python3
"""Interpret the inline-asm blocks tools/gen_mont_asm.py emits, on the CPU, instruction by instruction.

The generator writes straight-line gfx950 code over a handful of opcodes (v_mad_u64_u32 / v_mad_i64_i32, v_lshl_add_u64,
v_mul_lo_u32, v_and_b32, v_sub_u32, 64-bit shifts, v_mov_b32, v_lshlrev_b32, v_bfe_i32); this file gives each its ISA meaning
over Python integers, so a block can be checked against the big-integer Montgomery product here, before it is compiled
(tests/test_host_arith.py::test_generated_asm_blocks_compute_montgomery_products).  The 64-bit accumulator is v[0:1], as in
the blocks.

    python3 tools/sim_mont_asm.py        # shipped 9 / 10 / 14-limb blocks + the Karatsuba block of round 5, random operands
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_mont_asm import Block, KBlock  # noqa: E402

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def _sx32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def run_block(blk, values):
    """values: operand key ('a3', 'p0', 'ninv', ...) -> 32-bit integer for every input of the block; returns all operand values after
    the block ran (outputs under their keys, e.g. 'r0')"""
    n_out = len(blk.outs)
    names = {'%%%d' % (n if kind == 'o' else n_out + n): key for key, (kind, n) in blk.idx.items()}
    val = dict(values)
    acc = 0

    def get(tok):
        if tok == 'v[0:1]':
            return acc
        if tok == 'v0':
            return acc & M32
        if tok in names:
            return val[names[tok]]
        return int(tok, 0)

    for line in blk.lines:
        op, rest = line.split(' ', 1)
        a = [t.strip() for t in rest.split(',')]
        if op == 'v_mad_u64_u32':
            dst, res = a[0], ((get(a[2]) & M32) * (get(a[3]) & M32) + get(a[4])) & M64
        elif op == 'v_mad_i64_i32':
            dst, res = a[0], (_sx32(get(a[2])) * _sx32(get(a[3])) + get(a[4])) & M64
        elif op == 'v_lshl_add_u64':
            dst, res = a[0], ((get(a[1]) << get(a[2])) + get(a[3])) & M64
        elif op == 'v_sub_u32':
            dst, res = a[0], (get(a[1]) - get(a[2])) & M32
        elif op == 'v_mul_lo_u32':
            dst, res = a[0], (get(a[1]) * get(a[2])) & M32
        elif op == 'v_and_b32':
            dst, res = a[0], get(a[1]) & get(a[2])
        elif op == 'v_lshlrev_b32':
            dst, res = a[0], (get(a[2]) << get(a[1])) & M32
        elif op == 'v_lshrrev_b64':
            dst, res = a[0], get(a[2]) >> get(a[1])
        elif op == 'v_ashrrev_i64':
            v = get(a[2])
            v = v - (1 << 64) if v >> 63 else v
            dst, res = a[0], (v >> get(a[1])) & M64
        elif op == 'v_bfe_i32':
            w = get(a[3])
            f = (get(a[1]) >> get(a[2])) & ((1 << w) - 1)
            dst, res = a[0], (f - (1 << w) if f >> (w - 1) else f) & M32
        elif op == 'v_mov_b32':
            dst, res = a[0], get(a[1]) & M32
        else:
            raise ValueError('opcode the interpreter does not know: ' + op)
        if dst == 'v[0:1]':
            acc = res
        else:
            val[names[dst]] = res
    return val


def montgomery_check(blk, modulus, operands):
    """run an UNSIGNED block (kinds mul, sqr, mul2, kmul) on big-integer operands; returns (result, expected residue check)"""
    L, B = blk.L, blk.B
    mask = (1 << B) - 1
    limbs = lambda x: [(x >> (B * i)) & mask for i in range(L)]  # noqa: E731
    values = {'ninv': (-pow(modulus, -1, 1 << B)) % (1 << B)}
    for i, v in enumerate(limbs(modulus)):
        values['p%d' % i] = v
    for name, x in operands.items():
        for i, v in enumerate(limbs(x)):
            values['%s%d' % (name, i)] = v
    out = run_block(blk, values)
    r = [out['r%d' % i] for i in range(L)]
    if blk.kind == 'sqr':
        prod = operands['a'] ** 2
    elif blk.kind == 'mul2':
        prod = operands['a'] * operands['b'] + operands['c'] * operands['d']
    else:
        prod = operands['a'] * operands['b']
    value = sum(v << (B * i) for i, v in enumerate(r))
    ok = (value * (1 << (B * L)) - prod) % modulus == 0 and all(0 <= v <= mask for v in r[:-1])
    return r, ok


BLS_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
BN_Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def main():
    rng = random.Random(5)
    for L, modulus, bits in ((9, BLS_R, 257), (10, BN_Q, 262), (14, BLS_Q, 386)):
        for kind in ('mul', 'sqr', 'mul2'):
            blk = Block(L, 29, False, kind)
            blk.build()
            bound = bits - 1 if kind == 'mul2' else bits  # a b + c d must stay below 2^(29 L) p as well
            for _ in range(50):
                ops = {k: rng.randrange(1 << bound) for k in 'abcd'}
                assert montgomery_check(blk, modulus, ops)[1], (L, kind)
            print('MontAsm<%d>::%s: %d instructions, 50 random products right' % (L, kind, len(blk.lines)))
    k, u = KBlock(), Block(14, 29, False, 'mul')
    k.build()
    u.build()
    for t in range(200):
        ops = {'a': rng.randrange(1 << 386), 'b': rng.randrange(1 << 386)}
        if t == 0:
            ops = {'a': (1 << 386) - 1, 'b': (1 << 386) - 1}
        rk, ok = montgomery_check(k, BLS_Q, ops)
        ru, _ = montgomery_check(u, BLS_Q, ops)
        assert ok and rk == ru, t
    print('Karatsuba block (%d instructions, %d multiply-adds) == shipped block (%d instructions) limb for limb on 200 products'
          % (len(k.lines), k.mads, len(u.lines)))


if __name__ == '__main__':
    main()
