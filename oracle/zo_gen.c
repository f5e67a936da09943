/*
 * zo_gen.c -- CPU oracle: deterministic test-data generators.
 * TEST INFRASTRUCTURE ONLY (see zo.h).
 *
 * MT19937 restates vendor/mt.js/mt.es.js:84-152 (numeric seed path =
 * init_genrand; tempering as genrand_int32), and the three byte generators
 * restate test/browser.es/staff/util.js:87-134, so the inputs the reference's
 * own tests compress can be rebuilt from a seed.
 */
#include "zo.h"

void zo_mt_seed(zo_mt *s, uint32_t seed) { /* mt.es.js:86-92 */
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) {
        uint32_t x = s->mt[i - 1] ^ (s->mt[i - 1] >> 30);
        s->mt[i] = 1812433253u * x + (uint32_t)i;
    }
    s->idx = 624;
}

uint32_t zo_mt_next(zo_mt *s) { /* mt.es.js:125-152 */
    uint32_t *mt = s->mt, value;
    if (s->idx >= 624) {
        int k = 0;
        const int N = 624, M = 397;
        do {
            value = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            mt[k] = mt[k + M] ^ (value >> 1) ^ ((value & 1) ? 0x9908b0dfu : 0);
        } while (++k < N - M);
        do {
            value = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
            mt[k] = mt[k + M - N] ^ (value >> 1) ^ ((value & 1) ? 0x9908b0dfu : 0);
        } while (++k < N - 1);
        value = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[N - 1] = mt[M - 1] ^ (value >> 1) ^ ((value & 1) ? 0x9908b0dfu : 0);
        s->idx = 0;
    }
    value = mt[s->idx++];
    value ^= value >> 11;
    value ^= (value << 7) & 0x9d2c5680u;
    value ^= (value << 15) & 0xefc60000u;
    value ^= value >> 18;
    return value;
}

/* makeRandomData, util.js:87-97; nextInt(256) = low 8 bits (mt.es.js:172) */
void zo_gen_random(uint8_t *dst, size_t n, uint32_t seed) {
    zo_mt m;
    zo_mt_seed(&m, seed);
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)(zo_mt_next(&m) & 0xff);
}

/* makeSequentialData, util.js:102-111 */
void zo_gen_sequential(uint8_t *dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)(i & 0xff);
}

/* makeRandomSequentialData, util.js:116-134 */
void zo_gen_random_sequential(uint8_t *dst, size_t n, uint32_t seed) {
    zo_mt m;
    zo_mt_seed(&m, seed);
    for (size_t i = 0; i < n;) {
        uint32_t r1 = zo_mt_next(&m) & 0xff;
        uint32_t r2 = zo_mt_next(&m) & 0xff;
        while (r2--) {
            if (i == n) break;
            dst[i++] = (uint8_t)(r1++ & 0xff);
        }
    }
}

/* Zipf-like word text (this repo's synthetic corpus; no reference
 * counterpart).  A 4096-word vocabulary of random lowercase words (length
 * 2..10) is drawn with rank = r^3 >> 24 for a uniform 12-bit r, separated by
 * a space (newline after every 12th word).  Integer-only so that the JS twin
 * in tests/golden/gen_golden.mjs produces identical bytes. */
static inline uint32_t lcg(uint32_t *x) {
    *x = *x * 1664525u + 1013904223u;
    return *x;
}
void zo_gen_text(uint8_t *dst, size_t n, uint32_t seed) {
    static uint8_t vocab[4096][10];
    static uint8_t vlen[4096];
    uint32_t v = 0x9e3779b9u; /* the vocabulary is the same for every seed */
    for (int w = 0; w < 4096; ++w) {
        int len = 2 + (int)((lcg(&v) >> 16) % 9);
        vlen[w] = (uint8_t)len;
        for (int k = 0; k < len; ++k) vocab[w][k] = (uint8_t)('a' + (lcg(&v) >> 16) % 26);
    }
    uint32_t x = seed ^ 0x5bd1e995u;
    size_t i = 0;
    unsigned wc = 0;
    while (i < n) {
        uint64_t r = lcg(&x) >> 20; /* 12 bits */
        unsigned rank = (unsigned)((r * r * r) >> 24);
        for (int k = 0; k < vlen[rank] && i < n; ++k) dst[i++] = vocab[rank][k];
        if (i < n) dst[i++] = (++wc % 12 == 0) ? '\n' : ' ';
    }
}

/* nextInt(sup), mt.es.js:164-179 (min = 0) */
static uint32_t mt_next_int(zo_mt *m, uint32_t sup) {
    if ((sup & (~sup + 1)) == sup) return (sup - 1) & zo_mt_next(m);
    for (;;) {
        uint32_t value = zo_mt_next(m);
        uint32_t r = value % sup;
        if (!((uint64_t)sup > 4294967296ull - (uint64_t)(value - r))) return r;
    }
}

/* low-entropy runs: twin of genRuns in tests/golden/gen_golden.mjs */
void zo_gen_runs(uint8_t *dst, size_t n, uint32_t seed) {
    zo_mt m;
    zo_mt_seed(&m, seed);
    for (size_t i = 0; i < n;) {
        uint8_t b = (uint8_t)(mt_next_int(&m, 4) * 85);
        uint32_t run = 1 + mt_next_int(&m, 512);
        for (uint32_t k = 0; k < run && i < n; ++k) dst[i++] = b;
    }
}

/* k-letter alphabet: twin of 'small-alpha' in tests/golden/gen_golden.mjs */
void zo_gen_small_alpha(uint8_t *dst, size_t n, uint32_t seed, uint32_t k) {
    zo_mt m;
    zo_mt_seed(&m, seed);
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)(97 + mt_next_int(&m, k));
}
