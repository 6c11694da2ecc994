#!/usr/bin/env python3
"""CPU emulation of decode_mfma.hip's index arithmetic (no GPU needed): the tiled weight copy, the per-lane fragment addresses (full tiles and
PACKED half tiles: two k-steps per fragment), the 16 x 16 x 32 MFMA operand / accumulator maps, the K split over 8 waves and the reducer's
reads -- against a plain W @ x.  Run after touching any of those formulas:   python tools/mfma_layout_check.py"""
import numpy as np

rng = np.random.default_rng(0)
WAVES, R = 8, 8


def tile_weights(W, rows):   # dst[((b * (K / 32) + s) * rows + r) * 32 + c] = src[min(b * rows + r, N - 1)][32 s + c]
    N, K = W.shape
    out = np.zeros((256, K // 32, rows, 32), W.dtype)
    for b in range(256):
        for r in range(rows):
            out[b, :, r, :] = W[min(b * rows + r, N - 1)].reshape(K // 32, 32)
    return out


def lane_off(nrows, t, lane):            # mf_lane_off: (row, column group, k-step skip)
    return min(16 * t + (lane & 15), nrows - 1), lane >> 4, 0


def lane_off_half(nrows, first, skip, lane):   # mf_lane_off_half
    rho = lane & 15
    return min(first + (rho & 7), nrows - 1), lane >> 4, (rho >> 3) * skip


def frag(sl, s, off):                     # mf_frag: 8 bf16 of (row, 8 g .. 8 g + 7) at k-step s + skip
    row, g, skip = off
    return sl[s + skip, row, 8 * g:8 * g + 8]


def mfma(afrags, bfrags, acc):            # D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[l & 15][8 (l >> 4) ..], B[8 (l >> 4) ..][l & 15]
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = afrags[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = bfrags[l]
    return acc + A @ B                    # acc[i][j]: lane (j, g) holds rows 4 g .. 4 g + 3 of column j


def bfrag(xs, s, lane):                   # mf_bfrag: activation row (lane & 15) mod R, columns 32 s + 8 g ..
    return xs[(lane & 15) % R, 32 * s + 8 * (lane >> 4):32 * s + 8 * (lane >> 4) + 8]


def half_tile(lo, hi):                    # mf_half_tile: lane groups 0, 1 (rows 0..7) from lo, groups 2, 3 (rows 8..15) from hi
    out = lo.copy(); out[8:] = hi[8:]; return out


def skip_base(w, skip):                  # w2: wave w owns k-steps 11 w + i of the first column half and 88 + 11 w + i of the second
    return 11 * w


def run(N, K, rows, ksteps_per_wave, packed_first, skip, half_split=False):
    W = rng.standard_normal((N, K)); xs = rng.standard_normal((R, K))
    T = tile_weights(W, rows)
    for b in (0, 3, 255):
        sl = T[b]
        red = []                          # per wave: list of tiles
        for w in range(WAVES):
            tiles = []
            base = ksteps_per_wave * w if not half_split else skip_base(w, skip)
            if packed_first > 0:          # a full tile of rows 0..15 over all the wave's k-steps
                acc = np.zeros((16, 16))
                for i in range(ksteps_per_wave):
                    a = [frag(sl, base + i, lane_off(rows, 0, l)) for l in range(64)]
                    acc = mfma(a, [bfrag(xs, base + i, l) for l in range(64)], acc)
                tiles.append(acc)
            lo = np.zeros((16, 16)); hi = np.zeros((16, 16))
            for i in range(skip if not half_split else ksteps_per_wave // 2):   # packed half tile: fragment i = k-steps base + i | base + skip + i
                a = [frag(sl, base + i, lane_off_half(rows, packed_first, skip, l)) for l in range(64)]
                lo = mfma(a, [bfrag(xs, base + i, l) for l in range(64)], lo)
                hi = mfma(a, [bfrag(xs, base + skip + i, l) for l in range(64)], hi)
            tiles.append(half_tile(lo, hi))
            red.append(tiles)
        for i in range(rows):
            n = b * rows + i
            if n >= N: continue
            for j in range(R):
                if packed_first > 0 and i < 16: got = sum(red[w][0][i, j] for w in range(WAVES))
                else:
                    t = 1 if packed_first > 0 else 0
                    ii = i - packed_first
                    got = sum(red[w][t][ii, j] + red[w][t][ii + 8, j] for w in range(WAVES))
                want = W[n] @ xs[j]
                assert abs(got - want) < 1e-9 * (1 + abs(want)), (N, K, rows, b, i, j, got, want)
    print(f"ok: N={N} K={K} rows/workgroup={rows} k-steps/wave={ksteps_per_wave} packed rows from {packed_first}, pair distance {skip}")


run(6144, 2048, 24, 8, 16, 4)     # wqkv: rows 0..15 full, 16..23 packed (k-steps 8 w + i | 8 w + 4 + i)
run(2048, 2048, 8, 8, 0, 4)       # wo: 8 rows packed
run(5632, 2048, 22, 8, 16, 4)     # w1 / w3: rows 0..15 full, 16..21 packed (slots 6, 7 re-read row 21, never read back)
run(2048, 5632, 8, 22, 0, 88, half_split=True)   # w2: 8 rows packed, wave w: k-steps 11 w + i (first column half) | 88 + 11 w + i (second)
