#!/usr/bin/env python3
"""VERDICT r05 item 3 asked for SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS < 1 in k_mlp_rows (measured 2.1: 2.84 M conflict cycles over 1.34 M LDS
instructions per config-5 launch, profiles/r06/pmc_summary.json).  This is the bank model of MI355X_MICROARCH.md's LDS table applied to the
kernel's genre-row reads -- lane (r, q) of a wave reads the 16-byte piece q + 4 nb of sample r's row, sixteen different samples per lane group
of a ds_read_b128 (groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32; 64 banks of 4 bytes; an extra cycle per extra distinct address on a busy
bank) -- with ids drawn as bench.py draws them (uniform over 19, 2 % missing -> the shared zero row): the expected extra cycles per instruction
for the kernel's layout (rows 528 bytes apart) and for every other row stride / XOR swizzle of the 16-byte pieces.

Result (printed; profiles/r06/experiments/lds_conflict_model.txt): 5.4 extra cycles per genre-row read for the product's layout -- 8 192 tasks x
64 genre-row reads x 5.4 = 2.83 M: the counter's 2.84 M to the percent, i.e. EVERY conflict cycle of the kernel is a genre-row read (fragments
and vectors are conflict-free) -- and no stride or swizzle gets below 4.7 (544 bytes: -12 %): sixteen samples with sixteen
independent ids are sixteen random rows, and sixteen random 16-byte pieces into sixteen 4-bank groups collide like birthdays whatever the
map from (row, piece) to bank is.  Only identical ids broadcast.  A layout that is conflict-free for ANY ids needs the sixteen lanes of a group
on sixteen different piece indices, i.e. a per-lane rotation of the visiting order nb -> (nb + r) mod 8, which makes the accumulator index lane
dependent (a register index cannot be), or eight rotated copies of every table (640 KB)."""
import numpy as np

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def extra_cycles(addr):
    """addr[64]: byte address of each lane's 16-byte read.  Extra LDS cycles of one ds_read_b128."""
    extra = 0
    for g in GROUPS:
        per_bank = {}
        for l in g:
            a = int(addr[l])
            for w in range(4):                                    # the four dwords of the piece
                per_bank.setdefault(((a >> 2) + w) & 63, set()).add((a >> 2) + w)
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


def simulate(place, trials=4000, seed=1):
    rng = np.random.default_rng(seed)
    lane = np.arange(64)
    r, q = lane & 15, lane >> 4
    tot = 0
    for _ in range(trials):
        ids = rng.integers(0, 19, 16)
        ids[rng.random(16) < 0.02] = 19                           # the shared zero row
        nb = rng.integers(0, 8)
        tot += extra_cycles(np.array([place(int(ids[r[l]]), int(q[l]) + 4 * int(nb)) for l in range(64)]))
    return tot / trials


def main():
    print("row stride (bytes): extra LDS cycles per ds_read_b128 of a genre row (sixteen random ids per lane group)")
    best = None
    for stride in (512, 516, 520, 528, 544, 560, 576, 592, 640, 768):
        e = simulate(lambda i, p, s=stride: i * s + 16 * p)
        print("  %4d %s: %.2f" % (stride, "(product)" if stride == 528 else "         ", e))
        best = min(best or e, e) if stride != 512 else best
    for name, fn in (("512 + piece ^ (id & 31)", lambda i, p: i * 512 + 16 * (p ^ (i & 31))),
                     ("512 + piece ^ (id * 5 & 31)", lambda i, p: i * 512 + 16 * (p ^ ((i * 5) & 31))),
                     ("pieces transposed: [piece][id]", lambda i, p: (p * 20 + i) * 16)):
        e = simulate(fn)
        print("  %-32s: %.2f" % (name, e))
        best = min(best, e)
    print("  every id the same (broadcast)      : %.2f" % simulate(lambda i, p: 16 * p))
    print("best layout above: %.2f extra cycles; the product's 528-byte stride: %.2f" % (best, simulate(lambda i, p: i * 528 + 16 * p)))


if __name__ == "__main__":
    main()
