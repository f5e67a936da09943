// GPU round-trip verifier for raw DEFLATE streams (see include/zgpu_verify.h).
// One lane decodes one unit's stream bit-serially with canonical-code
// arithmetic (counts per length + symbols in code order, no lookup tables, so
// the per-lane state is ~0.8 KB) and checks it against the original bytes as it
// goes.  Strict where src/rawinflate.js is lenient (its LEN/NLEN test at :233
// is vacuous, it accepts incomplete code sets and never checks for leftovers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/zgpu_verify.h"

namespace {

struct Bits {
    const uint8_t *p, *end;
    uint64_t buf;
    uint32_t cnt;
    bool starved;
    __device__ void fill() {
        while (cnt <= 56 && p < end) {
            buf |= (uint64_t)(*p++) << cnt;
            cnt += 8;
        }
    }
    __device__ uint32_t take(uint32_t n) {  // n <= 16
        if (cnt < n) fill();
        if (cnt < n) {
            starved = true;
            return 0;
        }
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    }
};

struct Code {
    uint16_t count[16];  // codes per length
    uint16_t *symbol;    // symbols in canonical order
};

// lengths[0..n) -> canonical decoding arrays; returns Kraft slack (0 complete, <0 over-subscribed)
__device__ int build(Code &h, const uint8_t *lengths, int n) {
    for (int l = 0; l < 16; ++l) h.count[l] = 0;
    for (int s = 0; s < n; ++s) h.count[lengths[s]]++;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= h.count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
    for (int s = 0; s < n; ++s)
        if (lengths[s]) h.symbol[offs[lengths[s]]++] = (uint16_t)s;
    return left;
}

__device__ int decode(Bits &b, const Code &h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
        code |= (int)b.take(1);
        if (b.starved) return -1;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -2;
}

__constant__ uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                       35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513,
                                        769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
                                        9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// back: bytes of the same stream that precede orig (a chunk's matches may reach into them)
__device__ uint32_t verify_unit(const uint8_t *comp, uint64_t clen, const uint8_t *orig, uint64_t n, bool want_final,
                                uint64_t back) {
    Bits b{comp, comp + clen, 0, 0, false};
    uint16_t lsym[288], dsym[32];
    uint8_t lengths[320];
    Code lc, dc;
    lc.symbol = lsym;
    dc.symbol = dsym;
    uint64_t pos = 0;
    bool last = false;
    for (;;) {
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (b.starved) return ZGPU_V_TRUNCATED;
        if (type == 3) return ZGPU_V_BLOCK_TYPE;
        if (type == 0) {
            // stored: drop to the byte boundary, LEN, NLEN, bytes
            b.buf >>= (b.cnt & 7);
            b.cnt &= ~7u;
            const uint32_t len = b.take(16), nlen = b.take(16);
            if (b.starved) return ZGPU_V_TRUNCATED;
            if ((len ^ nlen) != 0xffffu) return ZGPU_V_STORED;
            if (pos + len > n) return ZGPU_V_LENGTH;
            for (uint32_t i = 0; i < len; ++i) {
                const uint32_t v = b.take(8);
                if (b.starved) return ZGPU_V_TRUNCATED;
                if (v != orig[pos + i]) return ZGPU_V_MISMATCH;
            }
            pos += len;
            if (last) break;
            // a chunk of a longer stream ends with an empty stored block and nothing after it
            if (!want_final && len == 0 && b.p == b.end && b.cnt == 0) break;
            continue;
        }
        if (type == 1) {
            for (int s = 0; s < 144; ++s) lengths[s] = 8;
            for (int s = 144; s < 256; ++s) lengths[s] = 9;
            for (int s = 256; s < 280; ++s) lengths[s] = 7;
            for (int s = 280; s < 288; ++s) lengths[s] = 8;
            build(lc, lengths, 288);
            for (int s = 0; s < 30; ++s) lengths[s] = 5;
            build(dc, lengths, 30);
        } else {
            const uint32_t nlen = b.take(5) + 257, ndist = b.take(5) + 1, ncode = b.take(4) + 4;
            if (b.starved) return ZGPU_V_TRUNCATED;
            if (nlen > 286 || ndist > 30) return ZGPU_V_HEADER;
            for (int i = 0; i < 19; ++i) lengths[i] = 0;
            for (uint32_t i = 0; i < ncode; ++i) lengths[kClOrder[i]] = (uint8_t)b.take(3);
            if (b.starved) return ZGPU_V_TRUNCATED;
            if (build(lc, lengths, 19) != 0) return ZGPU_V_CODESET;  // the code-length code must be complete
            uint32_t idx = 0;
            while (idx < nlen + ndist) {
                const int sym = decode(b, lc);
                if (sym < 0) return b.starved ? ZGPU_V_TRUNCATED : ZGPU_V_SYMBOL;
                if (sym < 16) {
                    lengths[idx++] = (uint8_t)sym;
                } else {
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) return ZGPU_V_HEADER;
                        val = lengths[idx - 1];
                        rep = 3 + b.take(2);
                    } else if (sym == 17)
                        rep = 3 + b.take(3);
                    else
                        rep = 11 + b.take(7);
                    if (b.starved) return ZGPU_V_TRUNCATED;
                    if (idx + rep > nlen + ndist) return ZGPU_V_HEADER;
                    while (rep--) lengths[idx++] = (uint8_t)val;
                }
            }
            if (lengths[256] == 0) return ZGPU_V_HEADER;
            int left = build(lc, lengths, (int)nlen);
            if (left < 0 || (left > 0 && nlen - lc.count[0] != 1)) return ZGPU_V_CODESET;  // one code of length 1 may stand alone
            left = build(dc, lengths + nlen, (int)ndist);
            if (left < 0 || (left > 0 && ndist - dc.count[0] > 1)) return ZGPU_V_CODESET;  // so may none at all (no matches)
        }
        // symbols until end of block
        for (;;) {
            const int sym = decode(b, lc);
            if (sym < 0) return b.starved ? ZGPU_V_TRUNCATED : ZGPU_V_SYMBOL;
            if (sym < 256) {
                if (pos >= n) return ZGPU_V_LENGTH;
                if ((uint32_t)sym != orig[pos]) return ZGPU_V_MISMATCH;
                ++pos;
            } else if (sym == 256) {
                break;
            } else {
                if (sym >= 286) return ZGPU_V_SYMBOL;
                const int li = sym - 257;
                const uint32_t len = kLenBase[li] + b.take(kLenExtra[li]);
                const int ds = decode(b, dc);
                if (ds < 0) return b.starved ? ZGPU_V_TRUNCATED : ZGPU_V_SYMBOL;
                if (ds >= 30) return ZGPU_V_SYMBOL;
                const uint32_t dist = kDistBase[ds] + b.take(kDistExtra[ds]);
                if (b.starved) return ZGPU_V_TRUNCATED;
                if (dist > pos + back) return ZGPU_V_DISTANCE;
                if (pos + len > n) return ZGPU_V_LENGTH;
                const uint8_t *src = orig + (int64_t)pos - (int64_t)dist;  // may lie before orig
                for (uint32_t i = 0; i < len; ++i)
                    if (orig[pos + i] != src[i]) return ZGPU_V_MISMATCH;
                pos += len;
            }
        }
        if (last) break;
    }
    if (pos != n) return ZGPU_V_LENGTH;
    // nothing may follow but the zero padding of the last byte
    if (b.p != b.end || b.cnt >= 8) return ZGPU_V_TRAILING;
    if (want_final != last) return ZGPU_V_TRAILING;
    return ZGPU_V_OK;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(64)
zgpu_verify_kernel(const uint8_t *comp, const uint64_t *comp_off, const uint8_t *orig, const uint64_t *orig_off,
                   uint32_t n_units, uint32_t flags, uint32_t *status) {
    const uint32_t u = blockIdx.x * 64 + threadIdx.x;
    if (u >= n_units) return;
    const bool want_final = (flags & 1u) || u == n_units - 1;
    const uint64_t back = (flags & 2u) ? orig_off[u] - orig_off[0] : 0;
    status[u] = verify_unit(comp + comp_off[u], comp_off[u + 1] - comp_off[u], orig + orig_off[u],
                            orig_off[u + 1] - orig_off[u], want_final, back);
}

extern "C" int zgpu_verify_streams(const void *d_comp, const uint64_t *d_comp_off, const void *d_orig,
                                   const uint64_t *d_orig_off, uint32_t n_units, uint32_t flags,
                                   uint32_t *d_status, void *stream) {
    if (!n_units) return 0;
    hipLaunchKernelGGL(zgpu_verify_kernel, dim3((n_units + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       (const uint8_t *)d_comp, d_comp_off, (const uint8_t *)d_orig, d_orig_off, n_units, flags,
                       d_status);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
