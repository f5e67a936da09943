/*
 * zo_checksum.c -- CPU oracle: CRC-32 and Adler-32 as the reference computes
 * them (src/crc32.js, src/adler32.js).  TEST INFRASTRUCTURE ONLY (see zo.h).
 */
#include "zo.h"

static uint32_t g_crc_table[256];
static int g_crc_ready;

/* The reference ships the table as literals (src/crc32.js:56-100); they are
 * the standard reflected table for polynomial 0xEDB88320, generated here. */
static void crc_table_init(void) {
    if (g_crc_ready) return;
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        g_crc_table[i] = c;
    }
    g_crc_ready = 1;
}

/* CRC32.update, src/crc32.js:25-42 (calc = update with crc 0, :14-16).
 * The reference consumes len&7 bytes first and then groups of eight; the
 * byte order is the same, so one loop states it. */
uint32_t zo_crc32_update(const uint8_t *data, size_t len, uint32_t crc) {
    crc_table_init();
    crc ^= 0xffffffffu;
    for (size_t i = 0; i < len; ++i) crc = (crc >> 8) ^ g_crc_table[(crc ^ data[i]) & 0xff];
    return crc ^ 0xffffffffu;
}

/* CRC32.single, src/crc32.js:48-50 */
uint32_t zo_crc32_single(uint32_t num, uint32_t crc) {
    crc_table_init();
    return g_crc_table[(num ^ crc) & 0xff] ^ (num >> 8);
}

/* Adler32.update, src/adler32.js:20-37 (mkHash = update(1, array), :11-13);
 * modulo taken every OptimizationParameter = 1024 bytes (:44). */
uint32_t zo_adler32_update(uint32_t adler, const uint8_t *data, size_t len) {
    uint32_t s1 = adler & 0xffff, s2 = (adler >> 16) & 0xffff;
    size_t i = 0;
    while (len > 0) {
        size_t tlen = len > 1024 ? 1024 : len;
        len -= tlen;
        do {
            s1 += data[i++];
            s2 += s1;
        } while (--tlen);
        s1 %= 65521;
        s2 %= 65521;
    }
    return (s2 << 16) | s1;
}
