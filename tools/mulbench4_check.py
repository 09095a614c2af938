#!/usr/bin/env python3
"""Verify the CHECK lines of tools/mulbench4 (stdin): r R = a b (mod p) for every variant's sample products."""
import sys
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
ok = bad = 0
for line in sys.stdin:
    if not line.startswith('CHECK'):
        continue
    t = line.split()
    name, L = t[1], int(t[2][2:])
    v = [int(x) for x in t[3:]]
    B = 29 if 'x29' in name else 30
    a, b, r = (sum(x << (B * i) for i, x in enumerate(v[j * L:(j + 1) * L])) for j in range(3))
    prod = 2 * a * b if name.endswith('mul2') else a * b
    good = (r * (1 << (B * L)) - prod) % P == 0
    lim = all(-(1 << (B - 1)) <= x < (1 << (B - 1)) for x in v[2 * L:3 * L - 1]) if B == 30 else all(0 <= x < (1 << B) for x in v[2 * L:3 * L - 1])
    ok += good and lim
    bad += not (good and lim)
    if not (good and lim):
        print('MISMATCH', name, good, lim)
print('mulbench4 check: %d ok, %d bad' % (ok, bad))
sys.exit(1 if bad or not ok else 0)
