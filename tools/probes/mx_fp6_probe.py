#!/usr/bin/env python
"""Analysis of tools/probes/mx_fp6_probe.hip's dump: which fp6 (e2m3) packing, scale semantics and k mapping the hardware uses.
    python tools/probes/mx_fp6_probe.py mx_fp6_probe.bin"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint8)
o = 0
def take(n, dt):
    global o
    a = raw[o:o + n * 4].view(dt).copy(); o += n * 4
    return a
A = take(32 * 64, np.float32).reshape(32, 64).astype(np.float64)
B = take(64 * 32, np.float32).reshape(64, 32).astype(np.float64)
ea, eb = take(64, np.int32), take(64, np.int32)
pa, pb = take(64 * 6, np.uint32).reshape(64, 6), take(64 * 6, np.uint32).reshape(64, 6)
D = take(32 * 32, np.float32).reshape(32, 32).astype(np.float64)

def e2m3_decode(code):
    s, e, m = (code >> 5) & 1, (code >> 3) & 3, code & 7
    v = np.where(e == 0, m / 8.0, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 1))
    return np.where(s == 1, -v, v)

def e2m3_quant(v):
    """round to nearest even onto the e2m3 grid, saturating at 7.5"""
    grid = np.array(sorted({float(e2m3_decode(np.array(c))) for c in range(32)}))
    a = np.minimum(np.abs(v), 7.5)
    idx = np.searchsorted(grid, a)
    idx = np.clip(idx, 1, len(grid) - 1)
    lo, hi = grid[idx - 1], grid[idx]
    pick_hi = (a - lo) > (hi - a)
    tie = (a - lo) == (hi - a)
    # ties to even mantissa: grid index parity (uniform steps within a binade)
    even_hi = (np.round(hi / (hi - lo)) % 2 == 0)
    q = np.where(pick_hi | (tie & even_hi), hi, lo)
    return np.sign(v) * q

def unpack(words):
    """[64 lanes][6 dwords] -> [64][32] 6-bit codes, little-endian bit stream"""
    bits = np.unpackbits(words.view(np.uint8).reshape(64, 24), axis=1, bitorder="little")
    codes = np.zeros((64, 32), np.int64)
    for b in range(6):
        codes += bits[:, b::6].astype(np.int64) << b
    return codes

ca, cb = unpack(pa), unpack(pb)
# expected values per lane: A lane l: row l%32, k = 32*(l//32)+0..31; B lane l: col l%32, same k
la = np.stack([A[l % 32, 32 * (l // 32):32 * (l // 32) + 32] for l in range(64)])
lb = np.stack([B[32 * (l // 32):32 * (l // 32) + 32, l % 32].astype(np.float16).astype(np.float64) for l in range(64)])
for name, codes, vals, e in (("A (2xpk16 from f32)", ca, la, ea), ("B (pk32 from f16)", cb, lb, eb)):
    dec = e2m3_decode(codes)
    for sem, sc in (("divide by scale", 2.0 ** (127.0 - e)), ("multiply by scale", 2.0 ** (e - 127.0))):
        want = e2m3_quant(vals * sc[:, None])
        bad = int((dec != want).sum())
        print(f"{name}: conversion semantics '{sem}': {bad} / {dec.size} codes differ")
# MFMA: D = sum_k 2^(ea-127) 2^(eb-127) deq(A) deq(B) with scale per (row, k-half)
qa = e2m3_decode(ca) * 2.0 ** (ea[:, None] - 127.0)      # [lane][32]
qb = e2m3_decode(cb) * 2.0 ** (eb[:, None] - 127.0)
Aq = np.zeros((32, 64)); Bq = np.zeros((64, 32))
for l in range(64):
    Aq[l % 32, 32 * (l // 32):32 * (l // 32) + 32] = qa[l]
    Bq[32 * (l // 32):32 * (l // 32) + 32, l % 32] = qb[l]
ref = Aq @ Bq
print("MFMA vs dequantised product (k = 32 h + e, scale = 2^(byte - 127) per lane):  max |D - ref| =", np.abs(D - ref).max(), " max |ref| =", np.abs(ref).max())
print("quantisation error of the product itself: max |ref - A B| / max|A B| =", np.abs(ref - A @ B).max() / np.abs(A @ B).max())
