"""CPU simulation behind DESIGN.md section 4 (arranged probe form): LDS-array cycles of the dense probe kernel's bucket reads
per B row and A-set.  A ds_read_b64 is served in two groups of 32 lanes; a group costs as many cycles as its most loaded
bank pair (= bucket mod 32) has distinct addresses.  Free form: lane l holds row elements l, l + 64, ...; arranged form:
pair_arrange_kernel's placement (lane = residue of the first bucket, NS slots per lane, overflow into any hole) and a
second bucket sharing `cbits` low bits with the first.  usage: probe_bank_cycles.py [s=1000] [trials=150]"""
import sys

import numpy as np

s = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(5)
buckets = 1
while buckets < s:
    buckets <<= 1
mask = np.uint32(buckets - 1)


def bucket2(x, cb):
    f = (x >> np.uint64(20)).astype(np.uint32) & mask
    if cb == 0 or int(mask) <= 2 * ((1 << cb) - 1):
        return f
    lm = (1 << cb) - 1
    c = (f & ~np.uint32(lm)) | (x.astype(np.uint32) & lm)
    return np.where(c == (x.astype(np.uint32) & mask), c ^ np.uint32(lm + 1), c)


def cycles(addresses):
    banks = {}
    for a in addresses:
        banks.setdefault(a % 32, set()).add(a)
    return max(len(v) for v in banks.values())


def row_pass(b1, b2, slots, ns):
    tot = 0
    for t in range(ns):
        for g in range(2):
            l1, l2 = [], []
            for lane in range(g * 32, g * 32 + 32):
                e = slots[t * 64 + lane]
                if e < 0:
                    l1.append(lane & 31 & int(mask)); l2.append(lane & 31 & int(mask))
                else:
                    l1.append(b1[e]); l2.append(b2[e])
            tot += cycles(l1) + cycles(l2)
    return tot


nt = (s + 63) // 64
print(f"s = {s}, {buckets} buckets, {trials} rows per form; LDS cycles per row and A-set (first + second bucket reads)")
for cb in (0, 2, 3, 4):
    for ns in sorted({nt, nt + (2 if nt > 4 else 1), nt + (4 if nt > 4 else 2)}):
        free = arr = mis = 0
        for _ in range(trials):
            x = rng.integers(0, 2**63, size=s, dtype=np.uint64) * 2 + rng.integers(0, 2, size=s, dtype=np.uint64)
            b1 = (x.astype(np.uint32) & mask).astype(int)
            slots = -np.ones(nt * 64, dtype=np.int64)
            slots[:s] = np.arange(s)
            free += row_pass(b1, bucket2(x, 0).astype(int), slots, nt)
            b2 = bucket2(x, cb).astype(int)
            slots = -np.ones(ns * 64, dtype=np.int64)
            cnt = np.zeros(32, dtype=int)
            over = []
            for e in range(s):
                r = b1[e] & 31
                k = cnt[r]; cnt[r] += 1
                if k < 2 * ns:
                    slots[(k >> 1) * 64 + (k & 1) * 32 + r] = e
                else:
                    over.append(e)
            at = 0
            for e in over:
                while slots[at] >= 0:
                    at += 1
                slots[at] = e
            mis += len(over)
            arr += row_pass(b1, b2, slots, ns)
        print(f"  cbits {cb}  NS {ns:2d}: arranged {arr / trials:6.1f}   free form {free / trials:6.1f}   hashes in a foreign lane {mis / trials:5.1f}")
