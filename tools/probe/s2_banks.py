"""LDS bank conflicts of the stride-2 weight-gradient kernel's transposed operand reads (ds_read_b64_tr_b16), brute force.
A lane (kq = lane >> 4, r = (lane & 15) >> 2, c4 = lane & 3) reads 8 bytes: patch pixel (2 kq + a, 2 (r + 4 s) + b), logical 32-byte half j,
channels 4 c4 .. +3.  The hardware serves 32 lanes (two kq row groups) per pass: 256 bytes = every one of the 64 banks once, if no two
lanes of a pass hit the same bank.  Layout parameters: row pitch in bytes, and which pixels have their two halves swapped."""
import itertools


def conflicts(pitch, flip):
    worst = 0
    for a, b, j, s in itertools.product(range(3), range(3), range(2), range(2)):
        for half in range(2):
            cnt = {}
            for lane in range(32 * half, 32 * half + 32):
                kq, r, c4 = lane >> 4, (lane & 15) >> 2, lane & 3
                py, px = 2 * kq + a, 2 * (r + 4 * s) + b
                addr = py * pitch + px * 64 + ((j ^ flip(py, px)) * 32) + c4 * 8
                for w in (addr // 4, addr // 4 + 1):
                    cnt[w % 64] = cnt.get(w % 64, 0) + 1
            worst = max(worst, max(cnt.values()))
    return worst


if __name__ == "__main__":
    flips = {"none": lambda py, px: 0, "px>>2": lambda py, px: (px >> 2) & 1, "px>>1": lambda py, px: (px >> 1) & 1,
             "px>>2 ^ py>>1": lambda py, px: ((px >> 2) ^ (py >> 1)) & 1, "px>>2 ^ py": lambda py, px: ((px >> 2) ^ py) & 1,
             "px>>1 ^ py>>1": lambda py, px: ((px >> 1) ^ (py >> 1)) & 1}
    for pitch in (17 * 64, 17 * 64 + 32, 18 * 64, 18 * 64 + 32, 17 * 64 + 16, 19 * 64, 20 * 64, 17 * 64 + 96):
        for name, f in flips.items():
            print(f"pitch {pitch:5d} B  flip {name:14s}: worst lanes per bank per pass = {conflicts(pitch, f)}")
