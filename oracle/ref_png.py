"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of the mask compressor of csrc/gsa_png.hip (include/gsa_png.h):
PNG filter "Up", run-length tokens coded with deflate's fixed Huffman code (RFC 1951 section 3.2.6), one block + an empty
stored block per group of 4 rows, Adler-32 of the filtered scanlines.  Small masks only (Python loops).  The stream is
pinned by zlib itself: ``zlib.decompress`` must return the filtered scanlines (tests/test_png.py)."""
import numpy as np

ROWS_PER_SEG = 4


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, bits, size):
        self.acc |= bits << self.n
        self.n += size
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, length):           # Huffman codes are packed starting from their most significant bit
        self.put(int(format(c, "0%db" % length)[::-1], 2), length)

    def literal(self, v):
        if v < 144:
            self.code(0x30 + v, 8)
        else:
            self.code(0x190 + v - 144, 9)

    def match1(self, length):
        ebits, extra = 0, 0
        if length == 258:
            sym = 285
        elif length <= 10:
            sym = 254 + length
        else:
            x = length - 3
            ebits = x.bit_length() - 1 - 2
            sym = 257 + 4 * (ebits + 1) + (x >> ebits) - 4
            extra = x & ((1 << ebits) - 1)
        if sym < 280:
            self.code(sym - 256, 7)
        else:
            self.code(0xC0 + sym - 280, 8)
        if ebits:
            self.put(extra, ebits)
        self.put(0, 5)

    def run(self, v, n):
        self.literal(v)
        n -= 1
        while n >= 3:
            m = min(n, 258)
            self.match1(m)
            n -= m
        for _ in range(n):
            self.literal(v)


def filtered_scanlines(mask):
    """PNG filter type 2 on every row: [2, row - row_above (mod 256)] -- what inflating the stream must give."""
    mask = np.asarray(mask, np.uint8)
    up = np.zeros_like(mask)
    up[1:] = mask[:-1]
    rows = np.empty((mask.shape[0], mask.shape[1] + 1), np.uint8)
    rows[:, 0] = 2
    rows[:, 1:] = mask - up
    return rows


def zlib_stream(mask):
    rows = filtered_scanlines(mask)
    H = rows.shape[0]
    out = bytearray(b"\x78\x01")
    for y0 in range(0, H, ROWS_PER_SEG):
        b = _Bits()
        b.put(2, 3)
        data = rows[y0:y0 + ROWS_PER_SEG].reshape(-1)
        # the kernel closes a run only when the value changes (runs continue across the rows of a group)
        i = 0
        while i < len(data):
            j = i
            while j < len(data) and data[j] == data[i]:
                j += 1
            b.run(int(data[i]), j - i)
            i = j
        b.put(0, 7)
        b.put(0, 3)
        if b.n:
            b.put(0, 8 - b.n)
        out += b.out + b"\x00\x00\xff\xff"
    out += b"\x03\x00"
    a, bsum = 1, 0
    for v in rows.reshape(-1).tolist():
        a = (a + v) % 65521
        bsum = (bsum + a) % 65521
    out += bytes([bsum >> 8, bsum & 255, a >> 8, a & 255])
    return bytes(out)
