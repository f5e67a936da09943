/*
 * zo_containers.c -- CPU oracle: the reference's zlib and gzip writers
 * (src/deflate.js:44-99, src/gzip.js:60-131) restated around zo_rawdeflate.
 * TEST INFRASTRUCTURE ONLY (see zo.h).
 */
#include "zo.h"
#include <stdlib.h>
#include <string.h>

/* Deflate.compress, src/deflate.js:44-99 */
size_t zo_zlib_compress(const uint8_t *in, size_t n, int ctype, int lazy, uint8_t *out,
                        size_t cap) {
    if (cap < 6) return (size_t)-1;
    if (ctype < 0 || ctype > 2) return (size_t)-1; /* 'unsupported compression type' :73 */
    size_t pos = 0;
    /* cinfo = LOG2E*log(0x8000) - 8 truncates to 7 under << (:52,:57) */
    uint32_t cmf = (7u << 4) | 8u;
    out[pos++] = (uint8_t)cmf;
    uint32_t flg = (uint32_t)ctype << 6; /* flevel = compressionType (:62-71), fdict 0 */
    uint32_t fcheck = 31 - ((cmf * 256 + flg) % 31);
    flg |= fcheck;
    out[pos++] = (uint8_t)flg;
    uint32_t adler = zo_adler32_update(1, in, n); /* :83 */
    size_t pos2;
    if (n == 0 && ctype == ZO_NONE) {
        /* RawDeflate.compress() hands back the whole untouched 0x8000-byte
         * output buffer for empty NONE input (rawdeflate.js:106,123), so the
         * trailer lands after 32768 bytes (:86-97). */
        if (cap < 0x8000 + 4) return (size_t)-1;
        memset(out + 2, 0, 0x8000 - 2);
        pos2 = 0x8000;
    } else {
        pos2 = zo_rawdeflate(in, n, ctype, lazy, out, cap, pos); /* :84-85 */
        if (pos2 == (size_t)-1 || pos2 + 4 > cap) return (size_t)-1;
    }
    out[pos2++] = (adler >> 24) & 0xff; /* :94-97 */
    out[pos2++] = (adler >> 16) & 0xff;
    out[pos2++] = (adler >> 8) & 0xff;
    out[pos2++] = adler & 0xff;
    return pos2;
}

/* Gzip.compress, src/gzip.js:60-131.  flags bit0 = flags.fname (sets FNAME
 * and writes the name, :71,:86), bit1 = flags.fcomment (sets FCOMMENT, :72),
 * bit2 = flags.fhcrc (:73,:102), bit3 = flags.comment (writes the comment
 * bytes, :94 -- the reference gates the bytes and the bit on different
 * option names). */
size_t zo_gzip_compress(const uint8_t *in, size_t n, int ctype, int lazy, uint32_t mtime,
                        int flags, const char *filename, const char *comment, uint8_t *out,
                        size_t cap) {
    size_t op = 0;
    size_t fl = filename ? strlen(filename) : 0, cl = comment ? strlen(comment) : 0;
    if (cap < 32 + fl + cl) return (size_t)-1;
    out[op++] = 0x1f;
    out[op++] = 0x8b;
    out[op++] = 8;
    uint8_t flg = 0;
    if (flags & 1) flg |= 0x08;
    if (flags & 2) flg |= 0x10;
    if (flags & 4) flg |= 0x02;
    out[op++] = flg;
    out[op++] = mtime & 0xff;
    out[op++] = (mtime >> 8) & 0xff;
    out[op++] = (mtime >> 16) & 0xff;
    out[op++] = (mtime >> 24) & 0xff;
    out[op++] = 0;
    out[op++] = 255; /* OperatingSystem.UNKNOWN :83 */
    if (flags & 1) {
        memcpy(out + op, filename, fl);
        op += fl;
        out[op++] = 0;
    }
    if (flags & 8) {
        memcpy(out + op, comment, cl);
        op += cl;
        out[op++] = 0;
    }
    if (flags & 4) {
        uint32_t crc16 = zo_crc32_update(out, op, 0) & 0xffff;
        out[op++] = crc16 & 0xff;
        out[op++] = (crc16 >> 8) & 0xff;
    }
    size_t op2 = zo_rawdeflate(in, n, ctype, lazy, out, cap, op); /* :107-111 */
    if (op2 == (size_t)-1 || op2 + 8 > cap) return (size_t)-1;
    uint32_t crc = zo_crc32_update(in, n, 0); /* :117 */
    out[op2++] = crc & 0xff;
    out[op2++] = (crc >> 8) & 0xff;
    out[op2++] = (crc >> 16) & 0xff;
    out[op2++] = (crc >> 24) & 0xff;
    uint32_t il = (uint32_t)n;
    out[op2++] = il & 0xff;
    out[op2++] = (il >> 8) & 0xff;
    out[op2++] = (il >> 16) & 0xff;
    out[op2++] = (il >> 24) & 0xff;
    return op2;
}
