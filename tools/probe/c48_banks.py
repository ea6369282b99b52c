"""LDS bank conflicts of the shared-stage 48 x 48 weight-gradient kernel (conv_wgrad_dma48.hip), brute force.
dy sub-tile: 4 rows x 8 pixels of 96 B (three 32-byte cout thirds); x patch: 6 rows x 10 pixels of 96 B (three cin thirds, one per compute wave).
A lane (kq = lane >> 4, r = (lane & 15) >> 2, c4 = lane & 3) of a ds_read_b64_tr_b16 reads 8 bytes of pixel (kq [+ta], r + 4 s [+tb]);
the hardware serves 32 lanes per pass."""
import itertools


def worst(addr_of):
    w = 0
    for half in range(2):
        cnt = {}
        for lane in range(32 * half, 32 * half + 32):
            a = addr_of(lane >> 4, (lane & 15) >> 2, lane & 3)
            for b in (a // 4, a // 4 + 1):
                cnt[b % 64] = cnt.get(b % 64, 0) + 1
        w = max(w, max(cnt.values()))
    return w


if __name__ == "__main__":
    for pitch in range(768, 1025, 32):
        ws = max(worst(lambda kq, r, c4: kq * pitch + (r + 4 * s) * 96 + i * 32 + c4 * 8) for i, s in itertools.product(range(3), range(2)))
        print(f"dy row pitch {pitch:4d} B: worst lanes per bank per pass = {ws}")
    for pitch in range(960, 1217, 32):
        ws = max(worst(lambda kq, r, c4: (kq + ta) * pitch + (r + 4 * s + tb) * 96 + w * 32 + c4 * 8)
                 for ta, tb, s, w in itertools.product(range(3), range(3), range(2), range(3)))
        print(f"x  row pitch {pitch:4d} B: worst lanes per bank per pass = {ws}")
