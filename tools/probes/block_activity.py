"""How many of k_conv_gather's gather instructions fetch row blocks without a single neighbour, and what coarser or finer skip
granularities would save: numpy model of the engine's position order (Morton order, rows sorted by neighbourhood mask inside
windows of 16384 positions) on synthetic 2 cm scenes (CPU only).
    python tools/probes/block_activity.py [scenes]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from languagegroundedsemseg_amd import synthetic


def masks_of(coords):
    c = coords.astype(np.int64)
    c[:, 1:] -= c[:, 1:].min(0) - 1
    key = ((c[:, 0] << 48) | (c[:, 1] << 32) | (c[:, 2] << 16) | c[:, 3])
    order = np.argsort(key)
    skey = key[order]
    m = np.zeros(len(c), np.uint32)
    for k in range(27):
        dx, dy, dz = k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1
        q = key + (dx << 32) + (dy << 16) + dz
        i = np.searchsorted(skey, q)
        i[i >= len(skey)] = 0
        m |= (skey[i] == q).astype(np.uint32) << k
    return m


def main():
    n_sc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    coords, _, _ = synthetic.make_batch(list(range(n_sc)))
    coords = coords[synthetic.morton_order(coords)]
    m = masks_of(coords)
    n = len(m)
    # windowed mask sort (k_mask_sort_keys, order 0)
    win = np.arange(n) // 16384
    perm = np.lexsort((m, win))
    m = m[perm]
    pad = (-n) % 256
    m = np.concatenate([m, np.zeros(pad, np.uint32)])
    bits = ((m[:, None] >> np.arange(27)[None]) & 1).astype(bool)          # [n_pad, 27]
    print("rows %d, neighbours per row %.2f" % (n, bits.sum() / n))
    tiles = bits.reshape(-1, 256, 27)
    visited = tiles.any(1)                                                   # [tiles, 27]
    for g in (32, 64, 128):
        blk = tiles.reshape(tiles.shape[0], 256 // g, g, 27).any(2)          # [tiles, blocks, 27]
        tot = visited.sum() * (256 // g)
        act = (blk & visited[:, None]).sum()
        print("granularity %3d rows: %.1f %% of the (visited tile-offset, block) pairs have a neighbour" % (g, 100.0 * act / tot))
    # rows with a neighbour inside active 32-row blocks (how uniform are the blocks)
    blk = tiles.reshape(tiles.shape[0], 8, 32, 27)
    a = blk.any(2)
    print("rows with a neighbour inside active 32-row blocks: %.1f %%" % (100.0 * blk.sum() / (a.sum() * 32)))
    # the wave owns blocks (2w, 2w+1): offsets per wave vs per tile
    w = tiles.reshape(tiles.shape[0], 4, 64, 27).any(2)
    print("offsets visited per tile %.2f, per 64-row wave %.2f, per 32-row block %.2f" % (visited.sum(1).mean(), w.sum(2).mean(), a.sum(2).mean()))


main()
