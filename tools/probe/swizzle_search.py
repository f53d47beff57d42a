#!/usr/bin/env python3
"""LDS swizzle keys for fragment reads of v_mfma_f32_16x16x32_f16 operands (conv_ht<m16>, conv_m16, gemm_x3, conv_gemm8<m16>).

A lane (l15 = lane & 15, kg = lane >> 4) reads the 16-byte chunk of k-group kg of row / pixel 16 blk + l15.  ds_read_b128 serves a
wave in four 16-lane groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table) -- so a
group mixes two k-groups, and the XOR key of the 32x32x16 kernels ((row >> 2) & 3 on 64-byte rows) puts two lanes of a group on one bank
quad.  This script searches, exhaustively over keys of a given period, for a key g(row) with chunk' = kg ^ g(row) that is conflict-free
for the tap shifts dx = 0, 1, 2 of a 3x3 halo image (64-byte rows), and checks the 128-byte-row case ((row >> 1) & 7, chunks 4 g + kg).
Host-only; no GPU needed.  Result used by the kernels: 64-byte rows: g(row) = ((row >> 2) & 1) << 1; 128-byte rows: (row >> 1) & 7."""
import itertools

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def conflict_free(chunk_of, row_bytes, shifts=(0, 1, 2), bases=(0, 16), nchunk_reads=(0,)):
    for base in bases:
        for dx in shifts:
            for g0 in nchunk_reads:
                for grp in GROUPS:
                    seen = set()
                    for lane in grp:
                        row = base + (lane & 15) + dx
                        addr = row * row_bytes + chunk_of(lane >> 4, row, g0) * 16
                        slot = (addr // 16) % 16          # 16 slots of 16 bytes = the 64 banks
                        if slot in seen:
                            return False
                        seen.add(slot)
    return True


def main():
    for period in (4, 8):
        hits = [v for v in itertools.product(range(4), repeat=period)
                if conflict_free(lambda kg, row, g0: kg ^ v[row % period], 64)]
        print(f"64-byte rows, key of period {period}: {len(hits)} conflict-free keys; first: {hits[:4]}")
    print("64-byte rows, the 32x32x16 kernels' key (row >> 2) & 3 under the 16x16x32 lane map:",
          conflict_free(lambda kg, row, g0: kg ^ ((row >> 2) & 3), 64))
    print("64-byte rows, ((row >> 2) & 1) << 1:", conflict_free(lambda kg, row, g0: kg ^ (((row >> 2) & 1) << 1), 64))
    print("128-byte rows, chunk 4 g + kg, key (row >> 1) & 7:",
          conflict_free(lambda kg, row, g0: (4 * g0 + kg) ^ ((row >> 1) & 7), 128, shifts=(0,), nchunk_reads=(0, 1)))


if __name__ == "__main__":
    main()


def stem_main():
    """`--stem`: 32-byte halo pixels (the fp16 space-to-depth stem of conv_hs.hip, 2 chunks per pixel, 32x32x16 lane map: lanes 0-31
    are rows r (lanes 0-15) and r + 1 (16-31) x 16 columns of one k-half).  Unswizzled, a service group's 16 lanes use 8 slots."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]

    def ok(pitch, g, taps=4):
        for dy in range(taps):
            for dx in range(taps):
                for r0 in (0, 2, 4, 6):
                    for ch in (0, 1):
                        for grp in groups:
                            seen = set()
                            for lane in grp:
                                row, col = r0 + (lane >> 4) + dy, (lane & 15) + dx
                                slot = (2 * (row * pitch + col) + (ch ^ g(row, col))) % 16
                                if slot in seen:
                                    return False
                                seen.add(slot)
        return True
    print("19-pixel rows, unswizzled:", ok(19, lambda r, c: 0), " chunk ^ (column & 1):", ok(19, lambda r, c: c & 1))


if __name__ == "__main__" and "--stem" in __import__("sys").argv:
    stem_main()
