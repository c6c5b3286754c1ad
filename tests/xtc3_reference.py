"""Independent restatement of the xtc3 coordinate compression for the tests (pure Python, slow).

Follows the published description of the format as implemented by the xdrfile library that GROMACS ships
(xdr3dfcoord: `sizeofint`, `sizeofints`, `sendbits`, `sendints` and the run loop), with its byte-array multi-precision
arithmetic, i.e. a different route to the bit string than the machine-word arithmetic of csrc/trajio.cpp. Used to pin
the C encoder's bytes on small frames; never imported by the package."""
import struct

MAGIC = [0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024,
         1290, 1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570,
         104031, 131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561, 832255, 1048576, 1321122, 1664510, 2097152, 2642245,
         3329021, 4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216]
FIRSTIDX = 9
LASTIDX = len(MAGIC)


class Bits:
    def __init__(self):
        self.out = bytearray(); self.lastbits = 0; self.lastbyte = 0

    def send(self, nbits, num):
        while nbits >= 8:
            self.lastbyte = ((self.lastbyte << 8) | ((num >> (nbits - 8)) & 0xFF)) & 0xFFFFFFFF
            self.out.append((self.lastbyte >> self.lastbits) & 0xFF)
            nbits -= 8
        if nbits > 0:
            self.lastbyte = ((self.lastbyte << nbits) | (num & ((1 << nbits) - 1))) & 0xFFFFFFFF
            self.lastbits += nbits
            if self.lastbits >= 8:
                self.lastbits -= 8
                self.out.append((self.lastbyte >> self.lastbits) & 0xFF)

    def finish(self):
        b = bytearray(self.out)
        if self.lastbits > 0:
            b.append((self.lastbyte << (8 - self.lastbits)) & 0xFF)
        return bytes(b)


def sizeofint(size):
    num, nbits = 1, 0
    while size >= num and nbits < 32:
        nbits += 1; num <<= 1
    return nbits


def _mul_bytes(by, size, add):
    tmp = add
    for k in range(len(by)):
        tmp = by[k] * size + tmp
        by[k] = tmp & 0xFF
        tmp >>= 8
    while tmp:
        by.append(tmp & 0xFF); tmp >>= 8


def sizeofints(sizes):
    by = [1]
    for s in sizes:
        _mul_bytes(by, s, 0)
    num, nbits = 1, 0
    top = by[-1]
    while top >= num:
        nbits += 1; num *= 2
    return nbits + (len(by) - 1) * 8


def sendints(bits, nbits, sizes, nums):
    by = []
    tmp = nums[0]
    while True:
        by.append(tmp & 0xFF); tmp >>= 8
        if tmp == 0:
            break
    for i in (1, 2):
        assert nums[i] < sizes[i]
        _mul_bytes(by, sizes[i], nums[i])
    if nbits >= len(by) * 8:
        for b in by:
            bits.send(8, b)
        bits.send(nbits - len(by) * 8, 0)
    else:
        for b in by[:-1]:
            bits.send(8, b)
        bits.send(nbits - (len(by) - 1) * 8, by[-1])


def _f32(v):
    return struct.unpack(">f", struct.pack(">f", v))[0]


def encode_frame(step, time, box9, x, precision=1000.0):
    """x: list of (x, y, z) floats. Returns the bytes of one .xtc frame."""
    n = len(x)
    out = bytearray(struct.pack(">iii", 1995, n, step) + struct.pack(">f", time))
    out += b"".join(struct.pack(">f", b) for b in box9)
    out += struct.pack(">i", n)
    if n <= 9:
        for p in x:
            out += struct.pack(">fff", *p)
        return bytes(out)
    out += struct.pack(">f", precision)
    prec = _f32(precision)
    ip = []
    minint, maxint = [2 ** 31 - 1] * 3, [-2 ** 31] * 3
    mindiff, old = 2 ** 31 - 1, [0, 0, 0]
    for a, p in enumerate(x):
        l = []
        for d in range(3):
            lf = _f32(_f32(p[d]) * prec)
            lf = _f32(lf + 0.5) if lf >= 0 else _f32(lf - 0.5)
            l.append(int(lf))
            minint[d] = min(minint[d], l[d]); maxint[d] = max(maxint[d], l[d])
        diff = sum(abs(o - c) for o, c in zip(old, l))
        if diff < mindiff and a > 0:
            mindiff = diff
        old = l
        ip.append(l)
    out += struct.pack(">iii", *minint) + struct.pack(">iii", *maxint)
    sizeint = [maxint[d] - minint[d] + 1 for d in range(3)]
    if (sizeint[0] | sizeint[1] | sizeint[2]) > 0xFFFFFF:
        bitsizeint = [sizeofint(s) for s in sizeint]; bitsize = 0
    else:
        bitsizeint = [0, 0, 0]; bitsize = sizeofints(sizeint)
    smallidx = FIRSTIDX
    while smallidx < LASTIDX - 1 and MAGIC[smallidx] < mindiff:
        smallidx += 1
    out += struct.pack(">i", smallidx)
    maxidx = min(LASTIDX - 1, smallidx + 8)
    minidx = maxidx - 8
    smaller = MAGIC[max(FIRSTIDX, smallidx - 1)] // 2
    smallnum = MAGIC[smallidx] // 2
    sizesmall = [MAGIC[smallidx]] * 3
    larger = MAGIC[maxidx] // 2
    bits = Bits()
    prev, prevrun, i = [0, 0, 0], -1, 0
    while i < n:
        is_small = False
        c = ip[i]
        if smallidx < maxidx and i >= 1 and all(abs(c[d] - prev[d]) < larger for d in range(3)):
            is_smaller = 1
        elif smallidx > minidx:
            is_smaller = -1
        else:
            is_smaller = 0
        if i + 1 < n and all(abs(c[d] - ip[i + 1][d]) < smallnum for d in range(3)):
            ip[i], ip[i + 1] = ip[i + 1], ip[i]
            c = ip[i]
            is_small = True
        tmp = [c[d] - minint[d] for d in range(3)]
        if bitsize == 0:
            for d in range(3):
                bits.send(bitsizeint[d], tmp[d])
        else:
            sendints(bits, bitsize, sizeint, tmp)
        prev = list(c)
        i += 1
        run = []
        if not is_small and is_smaller == -1:
            is_smaller = 0
        while is_small and len(run) < 24:
            c = ip[i]
            if is_smaller == -1 and sum((c[d] - prev[d]) ** 2 for d in range(3)) >= smaller * smaller:
                is_smaller = 0
            run += [c[d] - prev[d] + smallnum for d in range(3)]
            prev = list(c)
            i += 1
            is_small = i < n and all(abs(ip[i][d] - prev[d]) < smallnum for d in range(3))
        if len(run) != prevrun or is_smaller != 0:
            prevrun = len(run)
            bits.send(1, 1)
            bits.send(5, len(run) + is_smaller + 1)
        else:
            bits.send(1, 0)
        for k in range(0, len(run), 3):
            sendints(bits, smallidx, sizesmall, run[k:k + 3])
        if is_smaller != 0:
            smallidx += is_smaller
            if is_smaller < 0:
                smallnum = smaller; smaller = MAGIC[smallidx - 1] // 2
            else:
                smaller = smallnum; smallnum = MAGIC[smallidx] // 2
            sizesmall = [MAGIC[smallidx]] * 3
    blob = bits.finish()
    out += struct.pack(">i", len(blob)) + blob + b"\0" * (-len(blob) % 4)
    return bytes(out)
