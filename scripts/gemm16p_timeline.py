"""Reads the s_memtime stamps of the persistent GEMM (gemm16p_impl.h; scripts/probes/gemm16_bench.bin with MER_STAMP=<dir>) and prints
the per-tile cycle budget: medians over workgroups and steady-state tiles (tiles 2 .. of each workgroup)."""
import sys

import numpy as np



def main(path):
    a = np.fromfile(path, dtype=np.uint64).reshape(256, 2, 12, 16).astype(np.int64)
    for g in (0, 1):
        t = a[:, g]                                  # [wg, tile, slot]
        ok = (t[:, :, 0] > 0) & (t[:, :, 9] > 0)
        ntile = ok.sum(1)
        print(f"group {g}: workgroups with stamps {int((ntile > 0).sum())}, tiles per workgroup (stamped) {ntile.min()}..{ntile.max()}")
        steady = ok.copy()
        steady[:, :2] = False                        # skip the first two tiles of each workgroup
        steady &= np.roll(ok, -1, axis=1)            # ... and the last one (no next tile)
        steady[:, -1] = False
        if steady.sum() == 0:
            steady = ok
        d = lambda i, j: np.median((t[:, :, j] - t[:, :, i])[steady])
        rows = [("tile start -> mid kt0", 0, 1)] + [(f"mid kt{k} -> mid kt{k + 1}", 1 + k, 2 + k) for k in range(7)]
        rows += [("mid kt7 -> K loop done", 8, 9), ("K loop done -> boundary slab issued", 9, 10), ("boundary slab -> epilogue done", 10, 11)]
        if g == 1:
            rows = rows[:9] + [("mid kt7 -> K loop done", 8, 9), ("in-loop epilogue (13 -> 14)", 13, 14)]
        for name, i, j in rows:
            print(f"   {name:34s} {d(i, j):9.0f}")
        nxt = (np.roll(t[:, :, 0], -1, axis=1) - t[:, :, 0])[steady]
        print(f"   {'tile start -> next tile start':34s} {np.median(nxt):9.0f}   (p10 {np.percentile(nxt, 10):.0f}, p90 {np.percentile(nxt, 90):.0f})")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p)
        main(p)
