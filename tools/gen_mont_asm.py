#!/usr/bin/env python3
"""Emit hand-scheduled Montgomery products as ONE inline-asm block each (gfx950).

Why: in the C++ form of the product-scanning loop LLVM starts every column from 0 and adds the
incoming carry last (one v_lshl_add_u64 per column), and the instruction stream it schedules stalls a lone wave; neither can be
steered from the source without an asm boundary per term.  One block per product pins every column as a v_mad chain that starts
FROM the carry: 461 instead of 498 instructions for the 14 x 29-bit product, and -- measured, tools/mulbench4.hip,
profiles/r04_mulbench4_asm_vs_cpp_13x30.txt -- 794 instead of 938 ns per wave-product per SIMD at three waves per SIMD, at ANY
occupancy (the block issues back to back from a single wave).

The arithmetic is EXACTLY fu.hpp's C++ (same terms, same column order, same 64-bit wrap-around), so results are bit-identical
and the C++ stays the host path and the specification.

  --product   crypto3-zk_amd/csrc/mont_asm.hpp : unsigned B = 29 limbs, L in {9, 10, 14}: mul, sqr, mul2 (a b + c d), as
              specialisations MontAsm<L>
  --bench     tools/mont_asm.hpp : the forms tools/mulbench4.hip compares (signed centred 13 x 30-bit limbs with
              v_mad_i64_i32 -- measured only 2-5 % faster than the unsigned block, so not adopted; round 5: one Karatsuba
              level over the 14 x 29-bit product, KBlock)

The 64-bit column accumulator lives in the fixed pair v[0:1] (clobbered): inline-asm operands cannot name the halves of a
64-bit register operand, and the q digit needs the low half.  q_k shares its register with r_k (q_k is last read in column
k + L - 1, r_k is first written in column k + L).
"""
import argparse


class Block:
    def __init__(self, L, B, signed, kind):
        self.L, self.B, self.signed, self.kind = L, B, signed, kind
        self.outs, self.ins, self.idx = [], [], {}
        self.lines = []
        self.first = True

    def out(self, key, expr):
        self.idx[key] = ('o', len(self.outs))
        self.outs.append('"=&v"(%s)' % expr)

    def inp(self, key, expr, cons='v'):
        self.idx[key] = ('i', len(self.ins))
        self.ins.append('"%s"(%s)' % (cons, expr))

    def R(self, key):
        kind, n = self.idx[key]
        return '%%%d' % (n if kind == 'o' else len(self.outs) + n)

    def term(self, x, y):
        mad = 'v_mad_i64_i32' if self.signed else 'v_mad_u64_u32'
        src2 = '0' if self.first else 'v[0:1]'
        self.first = False
        self.lines.append('%s v[0:1], vcc, %s, %s, %s' % (mad, self.R(x), self.R(y), src2))

    def build(self):
        L, B, signed, kind = self.L, self.B, self.signed, self.kind
        shr = 'v_ashrrev_i64' if signed else 'v_lshrrev_b64'
        mask = (1 << B) - 1
        for i in range(L):
            self.out('r%d' % i, 'r[%d]' % i)
        if kind == 'sqr':
            for i in range(L):
                self.out('t%d' % i, 't[%d]' % i)
        for i in range(L):
            self.inp('a%d' % i, 'a[%d]' % i)
        if kind != 'sqr':
            for i in range(L):
                self.inp('b%d' % i, 'b[%d]' % i)
        if kind == 'mul2':
            for i in range(L):
                self.inp('c%d' % i, 'c[%d]' % i)
            for i in range(L):
                self.inp('d%d' % i, 'd[%d]' % i)
        for i in range(L):
            self.inp('p%d' % i, 'p[%d]' % i, 's')
        self.inp('ninv', 'ninv', 's')
        if signed:
            self.inp('bias', 'bias', 's')
        if kind == 'sqr':
            for i in range(L):
                self.lines.append('v_lshlrev_b32 %s, 1, %s' % (self.R('t%d' % i), self.R('a%d' % i)))
        for k in range(2 * L - 1):
            lo = 0 if k < L else k - L + 1
            hi = k if k < L else L - 1
            if kind == 'sqr':
                for i in range(lo, hi + 1):
                    j = k - i
                    if i < j:
                        self.term('a%d' % i, 't%d' % j)
                    elif i == j:
                        self.term('a%d' % i, 'a%d' % i)
            else:
                for i in range(lo, hi + 1):
                    self.term('a%d' % i, 'b%d' % (k - i))
                if kind == 'mul2':
                    for i in range(lo, hi + 1):
                        self.term('c%d' % i, 'd%d' % (k - i))
            if k < L:
                for i in range(k):
                    self.term('r%d' % i, 'p%d' % (k - i))
                q = self.R('r%d' % k)
                self.lines.append('v_mul_lo_u32 %s, v0, %s' % (q, self.R('ninv')))
                if signed:
                    self.lines.append('v_bfe_i32 %s, %s, 0, %d' % (q, q, B))
                else:
                    self.lines.append('v_and_b32 %s, 0x%x, %s' % (q, mask, q))
                self.term('r%d' % k, 'p0')
                self.lines.append('%s v[0:1], %d, v[0:1]' % (shr, B))
            else:
                for i in range(k - L + 1, L):
                    self.term('r%d' % i, 'p%d' % (k - i))
                r = self.R('r%d' % (k - L))
                if signed:
                    self.lines.append('v_bfe_i32 %s, v0, 0, %d' % (r, B))
                    self.lines.append('v_lshl_add_u64 v[0:1], v[0:1], 0, %s' % self.R('bias'))
                else:
                    self.lines.append('v_and_b32 %s, 0x%x, v0' % (r, mask))
                self.lines.append('%s v[0:1], %d, v[0:1]' % (shr, B))
        self.lines.append('v_mov_b32 %s, v0' % self.R('r%d' % (L - 1)))

    def emit(self, decl, indent='    '):
        T = 'int32_t' if self.signed else 'uint32_t'
        L = self.L
        args = ['%s (&r)[%d]' % (T, L), 'const %s (&a)[%d]' % (T, L)]
        if self.kind != 'sqr':
            args.append('const %s (&b)[%d]' % (T, L))
        if self.kind == 'mul2':
            args += ['const %s (&c)[%d]' % (T, L), 'const %s (&d)[%d]' % (T, L)]
        args += ['const %s (&p)[%d]' % (T, L), 'uint32_t ninv']
        o = [indent + '%s(%s) {' % (decl, ', '.join(args))]
        if self.kind == 'sqr':
            o.append(indent + '    %s t[%d];' % (T, L))
        if self.signed:
            o.append(indent + '    const uint64_t bias = 1ull << %d;' % (self.B - 1))
        o.append(indent + '    asm(')
        for ln in self.lines:
            o.append(indent + '        "%s\\n\\t"' % ln)
        o.append(indent + '        : %s' % ', '.join(self.outs))
        o.append(indent + '        : %s' % ', '.join(self.ins))
        o.append(indent + '        : "v0", "v1", "vcc");')
        o.append(indent + '}')
        return '\n'.join(o)


class KBlock(Block):
    """One subtractive Karatsuba level over 7 + 7 of the 14 x 29-bit limbs (tools/mulbench4.hip only).

    a b = P0 + X^7 (P0 + P2 + N) + X^14 P2 with P0 = a_lo b_lo, P2 = a_hi b_hi, N = (a_lo - a_hi)(b_hi - b_lo): with
    U[k] = P0[k] + P2[k - 7] (20 columns, 98 multiply-adds) column k of the product is U[k] + U[k - 7] + N[k - 7], so every held
    column is used twice with a PLUS sign.  147 multiply-adds instead of 196; the price is that the second use of a column sum
    cannot ride on a multiply-add's src2: 40 64-bit additions + 14 limb differences.  N's terms are signed (v_mad_i64_i32 into
    the same accumulator: exact mod 2^64, and the true column value is < 2^63).  The reduction half is the shipped one.
    U[k] lives in slot k mod 7 from column k to column k + 7.
    """

    def __init__(self):
        Block.__init__(self, 14, 29, False, 'kmul')

    def acc_term(self, mad, x, y):
        src2 = '0' if self.first else 'v[0:1]'
        self.first = False
        self.lines.append('%s v[0:1], vcc, %s, %s, %s' % (mad, self.R(x), self.R(y), src2))

    def acc_add(self, key):
        self.lines.append('v_lshl_add_u64 v[0:1], %s, 0, %s' % (self.R(key), '0' if self.first else 'v[0:1]'))
        self.first = False

    def build(self):
        L, B, H = 14, 29, 7
        mask = (1 << B) - 1
        for i in range(L):
            self.out('r%d' % i, 'r[%d]' % i)
        for i in range(H):
            self.out('da%d' % i, 'da[%d]' % i)
        for i in range(H):
            self.out('db%d' % i, 'db[%d]' % i)
        for i in range(H):
            self.out('u%d' % i, 'u[%d]' % i)
        for i in range(L):
            self.inp('a%d' % i, 'a[%d]' % i)
        for i in range(L):
            self.inp('b%d' % i, 'b[%d]' % i)
        for i in range(L):
            self.inp('p%d' % i, 'p[%d]' % i, 's')
        self.inp('ninv', 'ninv', 's')
        for i in range(H):
            self.lines.append('v_sub_u32 %s, %s, %s' % (self.R('da%d' % i), self.R('a%d' % i), self.R('a%d' % (i + H))))
            self.lines.append('v_sub_u32 %s, %s, %s' % (self.R('db%d' % i), self.R('b%d' % (i + H)), self.R('b%d' % i)))
        self.mads = 0
        for k in range(2 * L - 1):
            if H <= k <= 3 * H - 2:  # N[k - 7], seeded from the carry
                j = k - H
                for i in range(max(0, j - H + 1), min(H - 1, j) + 1):
                    self.acc_term('v_mad_i64_i32', 'da%d' % i, 'db%d' % (j - i))
                    self.mads += 1
            if k >= H:
                self.acc_add('u%d' % ((k - H) % H))
            if k <= 3 * H - 2:  # U[k] = P0[k] + P2[k - 7] into its slot, from 0
                slot = self.R('u%d' % (k % H))
                terms = []
                if k <= 2 * H - 2:
                    terms += [('a%d' % i, 'b%d' % (k - i)) for i in range(max(0, k - H + 1), min(H - 1, k) + 1)]
                j = k - H
                if 0 <= j <= 2 * H - 2:
                    terms += [('a%d' % (H + i), 'b%d' % (H + j - i)) for i in range(max(0, j - H + 1), min(H - 1, j) + 1)]
                for n, (x, y) in enumerate(terms):
                    self.lines.append('v_mad_u64_u32 %s, vcc, %s, %s, %s' % (slot, self.R(x), self.R(y), '0' if n == 0 else slot))
                    self.mads += 1
                self.acc_add('u%d' % (k % H))
            if k < L:
                for i in range(k):
                    self.term('r%d' % i, 'p%d' % (k - i))
                q = self.R('r%d' % k)
                self.lines.append('v_mul_lo_u32 %s, v0, %s' % (q, self.R('ninv')))
                self.lines.append('v_and_b32 %s, 0x%x, %s' % (q, mask, q))
                self.term('r%d' % k, 'p0')
                self.lines.append('v_lshrrev_b64 v[0:1], %d, v[0:1]' % B)
                self.mads += k + 1
            else:
                for i in range(k - L + 1, L):
                    self.term('r%d' % i, 'p%d' % (k - i))
                self.mads += 2 * L - 1 - k
                self.lines.append('v_and_b32 %s, 0x%x, v0' % (self.R('r%d' % (k - L)), mask))
                self.lines.append('v_lshrrev_b64 v[0:1], %d, v[0:1]' % B)
        self.lines.append('v_mov_b32 %s, v0' % self.R('r%d' % (L - 1)))

    def emit(self, decl, indent=''):
        o = [indent + '%s(uint32_t (&r)[14], const uint32_t (&a)[14], const uint32_t (&b)[14], const uint32_t (&p)[14], uint32_t ninv) {' % decl,
             indent + '    uint32_t da[7], db[7];', indent + '    uint64_t u[7];', indent + '    asm(']
        for ln in self.lines:
            o.append(indent + '        "%s\\n\\t"' % ln)
        o.append(indent + '        : %s' % ', '.join(self.outs))
        o.append(indent + '        : %s' % ', '.join(self.ins))
        o.append(indent + '        : "v0", "v1", "vcc");')
        o.append(indent + '}')
        return '\n'.join(o)


def product_header():
    parts = ['// generated by tools/gen_mont_asm.py --product -- do not edit (tests/test_abi.py checks it is up to date)',
             '// Montgomery products over unsigned 29-bit limbs as single inline-asm blocks; see the generator for why.',
             '#pragma once', '#include <cstdint>', '', 'namespace zkhip {', '',
             'template <int L>', 'struct MontAsm;  // L limbs of 29 bits', '']
    for L in (9, 10, 14):
        parts.append('template <>')
        parts.append('struct MontAsm<%d> {' % L)
        for kind in ('mul', 'sqr', 'mul2'):
            b = Block(L, 29, False, kind)
            b.build()
            parts.append('    // %d instructions' % len(b.lines))
            parts.append(b.emit('static __device__ __forceinline__ void %s' % kind))
        parts.append('};')
        parts.append('')
    parts.append('}  // namespace zkhip')
    return '\n'.join(parts) + '\n'


def bench_header():
    parts = ['// generated by tools/gen_mont_asm.py --bench -- do not edit', '#pragma once', '#include <cstdint>']
    for name, L, B, signed, kind in [('mont_s13_mul', 13, 30, True, 'mul'), ('mont_u14_mul', 14, 29, False, 'mul'),
                                     ('mont_s13_mul2', 13, 30, True, 'mul2')]:
        b = Block(L, B, signed, kind)
        b.build()
        parts.append('// %s: %d instructions' % (name, len(b.lines)))
        parts.append(b.emit('__device__ __forceinline__ void %s' % name, indent=''))
    k = KBlock()
    k.build()
    parts.append('// mont_k14_mul: %d instructions, %d multiply-adds (one Karatsuba level over the 14 x 29-bit product)' % (len(k.lines), k.mads))
    parts.append(k.emit('__device__ __forceinline__ void mont_k14_mul'))
    return '\n'.join(parts) + '\n'


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--product', help='write the product header (crypto3-zk_amd/csrc/mont_asm.hpp)')
    ap.add_argument('--bench', help='write the micro-benchmark header (tools/mont_asm.hpp)')
    a = ap.parse_args()
    if a.product:
        open(a.product, 'w').write(product_header())
    if a.bench:
        open(a.bench, 'w').write(bench_header())
