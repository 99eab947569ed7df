# brute-force an index skew p(idx) for the radix-4 Stockham passes: ds_write_b64 = 4 groups of 16 lanes, 16 slots (8 B)
# of 128 B; ds_read_b64 = 2 groups of 32 lanes, 32 slots of 256 B.
import itertools
def ways(slots, nslots):
    from collections import Counter
    c = Counter(s % nslots for s in set(slots))     # identical addresses broadcast
    return max(c.values())
def cost(p):
    tot = 0; worst = 0
    for Ns in (1, 4, 16, 64, 256):
        for r in range(4):
            for wave in range(4):
                # writes
                for grp in range(4):
                    lanes = [wave * 64 + grp * 16 + l for l in range(16)]
                    idx = [((t - (t & (Ns - 1))) << 2) + (t & (Ns - 1)) + r * Ns for t in lanes]
                    w = ways([p(i) for i in idx], 16); tot += w; worst = max(worst, w)
                for grp in range(2):
                    lanes = [wave * 64 + grp * 32 + l for l in range(32)]
                    idx = [t + 256 * r for t in lanes]
                    w = ways([p(i) for i in idx], 32); tot += w; worst = max(worst, w)
    return tot, worst
base = cost(lambda i: i)
print("identity", base)
best = []
for a, b, c in itertools.product(range(0, 8), range(0, 8), range(0, 8)):
    p = lambda i, a=a, b=b, c=c: i + a * (i >> 4) + b * (i >> 6) + c * (i >> 8)
    t, w = cost(p)
    best.append((t, w, a, b, c, p(1023)))
best.sort()
print(best[:8])
best2 = []
for a, b in itertools.product(range(0, 16), range(0, 16)):
    p = lambda i, a=a, b=b: i ^ (((i >> 4) * a) & 15) ^ (((i >> 8) * b) & 15)
    t, w = cost(p)
    best2.append((t, w, a, b))
best2.sort(); print(best2[:5])
for s1, s2 in itertools.product(range(2, 9), range(2, 9)):
    p = lambda i, s1=s1, s2=s2: i ^ ((i >> s1) & 15) ^ ((i >> s2) & 3)
    t, w = cost(p)
    if w <= 2: print("xor", s1, s2, t, w)
