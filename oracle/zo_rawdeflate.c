/*
 * zo_rawdeflate.c -- CPU oracle: restatement of the reference's raw DEFLATE
 * encoder (src/rawdeflate.js, src/bitstream.js, src/heap.js).
 *
 * TEST INFRASTRUCTURE ONLY (see zo.h).  The control flow follows the
 * JavaScript statement by statement, including its quirks (Uint16 heap
 * storage, 7-bit distance code limit, lazy>0 tail behaviour, dropped EOB on
 * empty input), because the byte-exact output of the reference is the parity
 * target.  Where JavaScript reads past the end of an array (yielding
 * `undefined`/NaN) the same outcome is modelled with NaN / -1 sentinels.
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* BitStream: src/bitstream.js:10-119                                  */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t *buf;
    size_t cap;
    size_t index;
    int bitindex;
    int overflow;
} zo_bs;

static uint8_t g_rev[256];
static int g_rev_ready;

/* buildReverseTable, src/bitstream.js:96-112 */
static void build_rev_table(void) {
    if (g_rev_ready) return;
    for (int i = 0; i < 256; ++i) {
        uint32_t n = (uint32_t)i, r = n;
        int s = 7;
        for (n >>= 1; n; n >>= 1) {
            r <<= 1;
            r |= n & 1;
            --s;
        }
        g_rev[i] = (uint8_t)((r << s) & 0xff);
    }
    g_rev_ready = 1;
}

/* rev32_, src/bitstream.js:74-81 */
static int32_t bs_rev32(uint32_t n) {
    return (int32_t)(((uint32_t)g_rev[n & 0xff] << 24) |
                     ((uint32_t)g_rev[(n >> 8) & 0xff] << 16) |
                     ((uint32_t)g_rev[(n >> 16) & 0xff] << 8) |
                     (uint32_t)g_rev[(n >> 24) & 0xff]);
}

static void bs_init(zo_bs *s, uint8_t *buf, size_t cap, size_t index) {
    build_rev_table();
    s->buf = buf;
    s->cap = cap;
    s->index = index;
    s->bitindex = 0;
    s->overflow = index >= cap;
}

/* writeBits, src/bitstream.js:42-67.  The reference grows its buffer on
 * demand (expandBuffer :30-35); here the caller supplies the worst-case
 * capacity and an overflow flag replaces growth. */
static void bs_write(zo_bs *s, uint32_t number, int n, int reverse) {
    if (s->overflow) return;
    size_t index = s->index;
    int bitindex = s->bitindex;
    uint32_t current = s->buf[index];
    if (reverse && n > 1) {
        if (n > 8)
            number = (uint32_t)(bs_rev32(number) >> (32 - n)); /* signed shift, as JS >> */
        else
            number = (uint32_t)(g_rev[number & 0xff] >> (8 - n));
    }
    if (n + bitindex < 8) {
        current = (current << n) | number;
        bitindex += n;
    } else {
        for (int i = 0; i < n; ++i) {
            current = (current << 1) | ((number >> (n - i - 1)) & 1);
            if (++bitindex == 8) {
                bitindex = 0;
                s->buf[index++] = g_rev[current & 0xff];
                current = 0;
                if (index == s->cap) {
                    s->overflow = 1;
                    s->index = index - 1;
                    return;
                }
            }
        }
    }
    s->buf[index] = (uint8_t)current;
    s->bitindex = bitindex;
    s->index = index;
}

/* finish, src/bitstream.js:86-95 */
static size_t bs_finish(zo_bs *s) {
    size_t index = s->index;
    if (s->overflow) return (size_t)-1;
    if (s->bitindex > 0) {
        s->buf[index] = (uint8_t)(s->buf[index] << (8 - s->bitindex));
        s->buf[index] = g_rev[s->buf[index]];
        index++;
    }
    return index;
}

/* ------------------------------------------------------------------ */
/* Lz77Match code tables: src/rawdeflate.js:624-772                    */
/* ------------------------------------------------------------------ */
/* Lz77Match.code(), :637-700 -> [code, extra, extraBits] */
static void length_code(int length, uint16_t out[3]) {
    static const uint16_t hi[] = {3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 22, 26, 30, 34,
                                  42, 50, 58, 66, 82, 98, 114, 130, 162, 194, 226, 257, 258};
    static const uint16_t base[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                    35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t eb[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int i = 0; i < 29; ++i)
        if (length <= hi[i]) {
            out[0] = (uint16_t)(257 + i);
            out[1] = (uint16_t)(length - base[i]);
            out[2] = eb[i];
            return;
        }
    abort(); /* 'invalid length' :698 */
}

/* getDistanceCode_, :707-772 -> [code, extra, extraBits] */
static void dist_code(int dist, uint16_t out[3]) {
    static const uint16_t hi[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256,
                                  384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192,
                                  12288, 16384, 24576, 32768};
    static const uint16_t base[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                    257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                    8193, 12289, 16385, 24577};
    static const uint8_t eb[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (int i = 0; i < 30; ++i)
        if (dist <= hi[i]) {
            out[0] = (uint16_t)i;
            out[1] = (uint16_t)(dist - base[i]);
            out[2] = eb[i];
            return;
        }
    abort(); /* 'invalid distance' :770 */
}

/* ------------------------------------------------------------------ */
/* lz77 + searchLongestMatch_: src/rawdeflate.js:283-395               */
/* ------------------------------------------------------------------ */
/* The reference keeps, per exact 3-byte key, a JS array of every earlier
 * position (:284,:319-327,:354).  Here the same candidate sets are walked
 * through hash chains; candidates whose three bytes differ are skipped, so
 * the set and its newest-to-oldest order are identical. */
#define ZO_HBITS 18
typedef struct {
    const uint8_t *d;
    size_t n;
    int32_t *head;
    int32_t *prev;
} zo_chain;

static inline uint32_t key3(const uint8_t *d, size_t n, size_t p) {
    /* :314-318 -- up to three bytes, big-endian; shorter at the very end */
    uint32_t k = 0;
    for (int i = 0; i < 3; ++i) {
        if (p + (size_t)i == n) break;
        k = (k << 8) | d[p + i];
    }
    return k;
}
static inline uint32_t hkey(uint32_t k) { return (k * 2654435761u) >> (32 - ZO_HBITS); }

typedef struct {
    uint16_t *buf;
    size_t cap; /* lz77buf.length = 2n, :287 */
    size_t pos;
    uint32_t *fl, *fd;
    long skip;
    int prev_valid, prev_len, prev_dist;
} zo_lz;

static void put_word(zo_lz *z, uint16_t w) {
    if (z->pos < z->cap) z->buf[z->pos] = w; /* writes past a typed array are ignored */
    z->pos++;
}

/* writeMatch, :303-311 (+ toLz77Array :779-791) */
static void write_match(zo_lz *z, int len, int dist, int offset) {
    uint16_t lc[3], dc[3];
    length_code(len, lc);
    dist_code(dist, dc);
    put_word(z, lc[0]);
    put_word(z, lc[1]);
    put_word(z, lc[2]);
    put_word(z, dc[0]);
    put_word(z, dc[1]);
    put_word(z, dc[2]);
    z->fl[lc[0]]++;
    z->fd[dc[0]]++;
    z->skip = len + offset - 1;
    z->prev_valid = 0;
}

#ifndef ZO_WINDOW /* diagnostic builds may shorten the window; the reference's is 32 KiB, :325-327 */
#define ZO_WINDOW 0x8000
#endif
size_t zo_lz77(const uint8_t *in, size_t n, int lazy, uint16_t *tokens, uint32_t *freqs_litlen,
               uint32_t *freqs_dist) {
    uint32_t fl_local[286], fd_local[30];
    zo_lz z;
    z.buf = tokens;
    z.cap = 2 * n;
    z.pos = 0;
    z.fl = freqs_litlen ? freqs_litlen : fl_local;
    z.fd = freqs_dist ? freqs_dist : fd_local;
    z.skip = 0;
    z.prev_valid = 0;
    z.prev_len = z.prev_dist = 0;
    memset(z.fl, 0, 286 * sizeof(uint32_t));
    memset(z.fd, 0, 30 * sizeof(uint32_t));
    z.fl[256] = 1; /* :296 */

    int32_t *head = (int32_t *)malloc(sizeof(int32_t) << ZO_HBITS);
    int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    memset(head, 0xff, sizeof(int32_t) << ZO_HBITS);

    for (size_t position = 0; position < n; ++position) {
        uint32_t k = key3(in, n, position);
        uint32_t h = hkey(k);
        if (z.skip-- > 0) { /* :321-324 */
            prev[position] = head[h];
            head[h] = (int32_t)position;
            continue;
        }
        if (position + 3 >= n) { /* :328-336 */
            if (z.prev_valid) write_match(&z, z.prev_len, z.prev_dist, -1);
            for (size_t i = 0, il = n - position; i < il; ++i) {
                uint8_t t = in[position + i];
                put_word(&z, t);
                ++z.fl[t];
            }
            break;
        }
        /* searchLongestMatch_, :370-395: newest to oldest, strictly longer
         * wins (ties keep the nearest), stop at 258; window :325-327. */
        int best_len = 0, best_dist = 0;
        size_t maxlen = n - position < 258 ? n - position : 258;
        for (int32_t c = head[h]; c >= 0 && position - (size_t)c <= ZO_WINDOW; c = prev[c]) {
            const uint8_t *a = in + c, *b = in + position;
            if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) continue; /* other key */
            if (best_len >= 3) {
                if ((size_t)best_len >= maxlen) break; /* nothing can be strictly longer */
                if (a[best_len] != b[best_len]) continue;
            }
            size_t l = 3;
            while (l < maxlen && a[l] == b[l]) ++l;
            if ((int)l > best_len) {
                best_len = (int)l;
                best_dist = (int)(position - (size_t)c);
            }
            if (l == 258) break;
        }
#ifdef ZO_MIN_MATCH /* diagnostic builds only: what the reference would make of a longer shortest match */
        if (best_len < ZO_MIN_MATCH) best_len = 0;
#endif
        if (best_len > 0) { /* :337-347 */
            if (z.prev_valid) {
                if (z.prev_len < best_len) {
                    uint8_t t = in[position - 1];
                    put_word(&z, t);
                    ++z.fl[t];
                    write_match(&z, best_len, best_dist, 0);
                } else
                    write_match(&z, z.prev_len, z.prev_dist, -1);
            } else if (best_len < lazy) {
                z.prev_valid = 1;
                z.prev_len = best_len;
                z.prev_dist = best_dist;
            } else
                write_match(&z, best_len, best_dist, 0);
        } else if (z.prev_valid) /* :348 */
            write_match(&z, z.prev_len, z.prev_dist, -1);
        else { /* :349-353 */
            uint8_t t = in[position];
            put_word(&z, t);
            ++z.fl[t];
        }
        prev[position] = head[h]; /* :354 */
        head[h] = (int32_t)position;
    }
    put_word(&z, 256); /* :356-357 */
    z.fl[256]++;
    free(head);
    free(prev);
    return z.pos < z.cap ? z.pos : z.cap; /* subarray(0,pos) clamps, :360 */
}

/* ------------------------------------------------------------------ */
/* Heap: src/heap.js:10-84                                             */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t buf[2 * 2 * 286 + 8];
    int length;
    uint32_t mask; /* 0xffff reproduces Uint16Array storage, heap.js:11 */
} zo_heap;

static int heap_parent(int index) { return ((index - 2) / 4) * 2; } /* :20-22 (index>=2) */

static void heap_push(zo_heap *h, uint32_t index, uint32_t value) { /* :37-55 */
    uint32_t *heap = h->buf;
    int current = h->length;
    heap[h->length++] = value & h->mask;
    heap[h->length++] = index & h->mask;
    while (current > 0) {
        int parent = heap_parent(current);
        if (heap[current] > heap[parent]) {
            uint32_t s1 = heap[current];
            heap[current] = heap[parent];
            heap[parent] = s1;
            uint32_t s2 = heap[current + 1];
            heap[current + 1] = heap[parent + 1];
            heap[parent + 1] = s2;
            current = parent;
        } else
            break;
    }
}

static void heap_pop(zo_heap *h, uint32_t *index, uint32_t *value) { /* :61-84 */
    uint32_t *heap = h->buf;
    *value = heap[0];
    *index = heap[1];
    h->length -= 2;
    heap[0] = heap[h->length];
    heap[1] = heap[h->length + 1];
    int parent = 0;
    for (;;) {
        int current = 2 * parent + 2;
        if (current >= h->length) break;
        if (current + 2 < h->length && heap[current + 2] > heap[current]) current += 2;
        if (heap[current] > heap[parent]) {
            uint32_t s1 = heap[parent];
            heap[parent] = heap[current];
            heap[current] = s1;
            uint32_t s2 = heap[parent + 1];
            heap[parent + 1] = heap[current + 1];
            heap[current + 1] = s2;
        } else
            break;
        parent = current;
    }
}

/* ------------------------------------------------------------------ */
/* reversePackageMerge_ + takePackage: src/rawdeflate.js:497-574       */
/* ------------------------------------------------------------------ */
#define RPM_MAXLIMIT 16
typedef struct {
    int limit, symbols;
    double *value[RPM_MAXLIMIT];
    int *type[RPM_MAXLIMIT]; /* -1 = undefined */
    int width;               /* allocated entries per level */
    int cur[RPM_MAXLIMIT];
    uint8_t *code_length;
} zo_rpm;

static double rpm_val(const zo_rpm *r, int j, int t) {
    return (t >= 0 && t < r->width) ? r->value[j][t] : NAN;
}
static int rpm_type(const zo_rpm *r, int j, int t) {
    return (t >= 0 && t < r->width) ? r->type[j][t] : -1;
}

static void take_package(zo_rpm *r, int j) { /* :497-504 */
    if (j >= r->limit) abort(); /* JS would throw a TypeError here */
    int x = rpm_type(r, j, r->cur[j]);
    if (x == r->symbols) {
        take_package(r, j + 1);
        take_package(r, j + 1);
    } else if (x >= 0 && x < r->symbols)
        --r->code_length[x];
    /* x undefined: --codeLength[undefined] touches no element */
    ++r->cur[j];
}

static void reverse_package_merge(const uint32_t *freqs, int symbols, int limit,
                                  uint8_t *code_length) { /* :512-574 */
    zo_rpm r;
    int limitM1 = limit - 1;
    uint16_t minimum_cost[RPM_MAXLIMIT];
    uint8_t flag[RPM_MAXLIMIT];
    memset(minimum_cost, 0, sizeof minimum_cost);
    memset(flag, 0, sizeof flag);
    r.limit = limit;
    r.symbols = symbols;
    r.code_length = code_length;
    r.width = 2 * symbols + 8;
    for (int j = 0; j < limit; ++j) {
        r.value[j] = (double *)malloc(sizeof(double) * (size_t)r.width);
        r.type[j] = (int *)malloc(sizeof(int) * (size_t)r.width);
        for (int t = 0; t < r.width; ++t) {
            r.value[j][t] = NAN;
            r.type[j][t] = -1;
        }
    }
    int32_t excess = (int32_t)((1u << limit) - (uint32_t)symbols);
    int32_t half = 1 << limitM1;
    minimum_cost[limitM1] = (uint16_t)symbols;
    for (int j = 0; j < limit; ++j) {
        if (excess < half)
            flag[j] = 0;
        else {
            flag[j] = 1;
            excess -= half;
        }
        excess <<= 1;
        if (limit - 2 - j >= 0)
            minimum_cost[limit - 2 - j] = (uint16_t)((minimum_cost[limitM1 - j] / 2) + symbols);
    }
    minimum_cost[0] = flag[0];
    for (int j = 1; j < limit; ++j)
        if (minimum_cost[j] > 2 * minimum_cost[j - 1] + flag[j])
            minimum_cost[j] = (uint16_t)(2 * minimum_cost[j - 1] + flag[j]);
    for (int i = 0; i < symbols; ++i) code_length[i] = (uint8_t)limit;
    int tml = minimum_cost[limitM1];
    for (int t = 0; t < tml; ++t) {
        r.value[limitM1][t] = (t < symbols) ? (double)freqs[t] : NAN;
        r.type[limitM1][t] = t;
    }
    for (int i = 0; i < limit; ++i) r.cur[i] = 0;
    if (flag[limitM1] == 1) {
        if (symbols > 0) --code_length[0];
        ++r.cur[limitM1];
    }
    for (int j = limit - 2; j >= 0; --j) {
        int i = 0;
        int next = r.cur[j + 1];
        int mcj = minimum_cost[j];
        for (int t = 0; t < mcj; t++) {
            double weight = rpm_val(&r, j + 1, next) + rpm_val(&r, j + 1, next + 1);
            double fi = (i < symbols) ? (double)freqs[i] : NAN;
            if (weight > fi) {
                r.value[j][t] = weight;
                r.type[j][t] = symbols;
                next += 2;
            } else {
                r.value[j][t] = fi;
                r.type[j][t] = i;
                ++i;
            }
        }
        r.cur[j] = 0;
        if (flag[j] == 1) take_package(&r, j);
    }
    for (int j = 0; j < limit; ++j) {
        free(r.value[j]);
        free(r.type[j]);
    }
}

/* getLengths_: src/rawdeflate.js:474-493 */
void zo_get_lengths(const uint32_t *freqs, int nsym, int limit, int heap16, uint8_t *lengths) {
    zo_heap heap;
    uint32_t node_index[286], values[286];
    heap.length = 0;
    heap.mask = heap16 ? 0xffffu : 0xffffffffu;
    memset(lengths, 0, (size_t)nsym);
    for (int i = 0; i < nsym; ++i)
        if (freqs[i] > 0) heap_push(&heap, (uint32_t)i, freqs[i]);
    int cnt = heap.length / 2;
    if (cnt == 1) {
        uint32_t idx, val;
        heap_pop(&heap, &idx, &val);
        lengths[idx] = 1;
        return;
    }
    for (int i = 0; i < cnt; ++i) heap_pop(&heap, &node_index[i], &values[i]);
    uint8_t code_length[286];
    reverse_package_merge(values, cnt, limit, code_length);
    for (int i = 0; i < cnt; ++i) lengths[node_index[i]] = code_length[i];
}

/* getCodesFromLengths_: src/rawdeflate.js:582-605 */
void zo_get_codes(const uint8_t *lengths, int nsym, uint16_t *codes) {
    uint32_t count[256], start_code[17];
    memset(count, 0, sizeof count);
    uint32_t code = 0;
    for (int i = 0; i < nsym; ++i) count[lengths[i]]++;
    for (int i = 1; i <= 16; i++) {
        start_code[i] = code;
        code += count[i];
        code <<= 1;
    }
    for (int i = 0; i < nsym; i++) {
        int len = lengths[i];
        uint16_t out = 0;
        if (len >= 1 && len <= 16) {
            uint32_t c = start_code[len];
            start_code[len] += 1;
            for (int j = 0; j < len; j++) {
                out = (uint16_t)((out << 1) | (c & 1));
                c >>= 1;
            }
        }
        codes[i] = out;
    }
}

/* getTreeSymbols_: src/rawdeflate.js:408-466 */
static int get_tree_symbols(int hlit, const uint8_t *ll, int hdist, const uint8_t *dl,
                            uint32_t *result, uint8_t freqs[19]) {
    uint32_t src[286 + 30];
    int src_len = hlit + hdist, j = 0, n_result = 0;
    memset(freqs, 0, 19);
    for (int i = 0; i < hlit; i++) src[j++] = ll[i];
    for (int i = 0; i < hdist; i++) src[j++] = dl[i];
    for (int i = 0; i < src_len; i += j) {
        uint32_t s = src[i];
        for (j = 1; i + j < src_len && src[i + j] == s;) ++j;
        int run = j;
        if (s == 0) {
            if (run < 3)
                while (run-- > 0) {
                    result[n_result++] = 0;
                    freqs[0]++;
                }
            else
                while (run > 0) {
                    int rpt = run < 138 ? run : 138;
                    if (rpt > run - 3 && rpt < run) rpt = run - 3;
                    if (rpt <= 10) {
                        result[n_result++] = 17;
                        result[n_result++] = (uint32_t)(rpt - 3);
                        freqs[17]++;
                    } else {
                        result[n_result++] = 18;
                        result[n_result++] = (uint32_t)(rpt - 11);
                        freqs[18]++;
                    }
                    run -= rpt;
                }
        } else {
            result[n_result++] = s;
            freqs[s]++;
            run--;
            if (run < 3)
                while (run-- > 0) {
                    result[n_result++] = s;
                    freqs[s]++;
                }
            else
                while (run > 0) {
                    int rpt = run < 6 ? run : 6;
                    if (rpt > run - 3 && rpt < run) rpt = run - 3;
                    result[n_result++] = 16;
                    result[n_result++] = (uint32_t)(rpt - 3);
                    freqs[16]++;
                    run -= rpt;
                }
        }
    }
    return n_result;
}

/* ------------------------------------------------------------------ */
/* block makers + compress: src/rawdeflate.js:102-277                  */
/* ------------------------------------------------------------------ */
/* FixedHuffmanTable, :48-68 -> [code, bits] */
static void fixed_code(int i, uint32_t *code, int *bits) {
    if (i <= 143) {
        *code = (uint32_t)(i + 0x030);
        *bits = 8;
    } else if (i <= 255) {
        *code = (uint32_t)(i - 144 + 0x190);
        *bits = 9;
    } else if (i <= 279) {
        *code = (uint32_t)(i - 256);
        *bits = 7;
    } else {
        *code = (uint32_t)(i - 280 + 0x0c0);
        *bits = 8;
    }
}

/* makeFixedHuffmanBlock :160-169 + fixedHuffman :266-277 */
static size_t make_fixed_block(const uint8_t *in, size_t n, int lazy, uint8_t *out, size_t cap,
                               size_t op, int is_final) {
    zo_bs s;
    bs_init(&s, out, cap, op);
    bs_write(&s, is_final ? 1 : 0, 1, 1);
    bs_write(&s, ZO_FIXED, 2, 1);
    uint16_t *tok = (uint16_t *)malloc(sizeof(uint16_t) * (2 * n + 8));
    size_t nt = zo_lz77(in, n, lazy, tok, NULL, NULL);
    for (size_t i = 0; i < nt; i++) {
        uint16_t literal = tok[i];
        uint32_t code;
        int bits;
        fixed_code(literal, &code, &bits);
        bs_write(&s, code, bits, 0);
        if (literal > 0x100) {
            bs_write(&s, tok[i + 1], tok[i + 2], 1);
            bs_write(&s, tok[i + 3], 5, 0);
            bs_write(&s, tok[i + 4], tok[i + 5], 1);
            i += 5;
        } else if (literal == 0x100)
            break;
    }
    free(tok);
    return bs_finish(&s);
}

/* makeDynamicHuffmanBlock :176-236 + dynamicHuffman :243-259 */
static size_t make_dynamic_block(const uint8_t *in, size_t n, int lazy, int heap16, uint8_t *out,
                                 size_t cap, size_t op, int is_final) {
    static const uint8_t hclen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
                                            11, 4, 12, 3, 13, 2, 14, 1, 15};
    zo_bs s;
    bs_init(&s, out, cap, op);
    bs_write(&s, is_final ? 1 : 0, 1, 1);
    bs_write(&s, ZO_DYNAMIC, 2, 1);
    uint16_t *tok = (uint16_t *)malloc(sizeof(uint16_t) * (2 * n + 8));
    uint32_t fl[286], fd[30];
    size_t nt = zo_lz77(in, n, lazy, tok, fl, fd);
    uint8_t ll[286], dl[30], tl[19], trans[19];
    uint16_t lc[286], dc[30], tc[19];
    zo_get_lengths(fl, 286, 15, heap16, ll);
    zo_get_codes(ll, 286, lc);
    zo_get_lengths(fd, 30, 7, heap16, dl);
    zo_get_codes(dl, 30, dc);
    int hlit, hdist, hclen;
    for (hlit = 286; hlit > 257 && ll[hlit - 1] == 0;) hlit--;
    for (hdist = 30; hdist > 1 && dl[hdist - 1] == 0;) hdist--;
    uint32_t tree[286 + 30 + 8];
    uint8_t tfreq8[19];
    int ntree = get_tree_symbols(hlit, ll, hdist, dl, tree, tfreq8);
    uint32_t tfreq[19];
    for (int i = 0; i < 19; i++) tfreq[i] = tfreq8[i]; /* freqs is a Uint8Array, :412 */
    zo_get_lengths(tfreq, 19, 7, heap16, tl);
    for (int i = 0; i < 19; i++) trans[i] = tl[hclen_order[i]];
    for (hclen = 19; hclen > 4 && trans[hclen - 1] == 0;) hclen--;
    zo_get_codes(tl, 19, tc);
    bs_write(&s, (uint32_t)(hlit - 257), 5, 1);
    bs_write(&s, (uint32_t)(hdist - 1), 5, 1);
    bs_write(&s, (uint32_t)(hclen - 4), 4, 1);
    for (int i = 0; i < hclen; i++) bs_write(&s, trans[i], 3, 1);
    for (int i = 0; i < ntree; i++) {
        uint32_t code = tree[i];
        bs_write(&s, tc[code], tl[code], 1);
        if (code >= 16) {
            i++;
            int bitlen = code == 16 ? 2 : code == 17 ? 3 : 7;
            bs_write(&s, tree[i], bitlen, 1);
        }
    }
    for (size_t i = 0; i < nt; ++i) {
        uint16_t literal = tok[i];
        bs_write(&s, lc[literal], ll[literal], 1);
        if (literal > 256) {
            bs_write(&s, tok[i + 1], tok[i + 2], 1);
            uint16_t code = tok[i + 3];
            bs_write(&s, dc[code], dl[code], 1);
            bs_write(&s, tok[i + 4], tok[i + 5], 1);
            i += 5;
        } else if (literal == 256)
            break;
    }
    free(tok);
    return bs_finish(&s);
}

size_t zo_deflate_bound(size_t n) { return 2 * n + 5 * (n / 65535 + 1) + 1024; }

static size_t rawdeflate_impl(const uint8_t *in, size_t n, int ctype, int lazy, int heap16,
                              uint8_t *out, size_t cap, size_t out_index) {
    if (cap <= out_index) return (size_t)-1;
    memset(out + out_index, 0, cap - out_index); /* fresh Uint8Array is zero-filled */
    switch (ctype) {
    case ZO_NONE: { /* :105-111 + makeNocompressBlock :131-153 */
        size_t op = out_index;
        for (size_t position = 0; position < n;) {
            size_t blen = n - position < 0xffff ? n - position : 0xffff;
            position += blen;
            if (op + blen + 5 > cap) return (size_t)-1;
            out[op++] = (uint8_t)((position == n ? 1 : 0) | (ZO_NONE << 1));
            uint32_t len = (uint32_t)blen, nlen = (~len + 0x10000u) & 0xffffu;
            out[op++] = len & 0xff;
            out[op++] = (len >> 8) & 0xff;
            out[op++] = nlen & 0xff;
            out[op++] = (nlen >> 8) & 0xff;
            memcpy(out + op, in + position - blen, blen);
            op += blen;
        }
        /* empty input: the loop body never runs and compress() returns the
         * untouched output buffer (:106,:123); callers see out_index bytes. */
        return op;
    }
    case ZO_FIXED:
        return make_fixed_block(in, n, lazy, out, cap, out_index, 1);
    case ZO_DYNAMIC:
        return make_dynamic_block(in, n, lazy, heap16, out, cap, out_index, 1);
    default:
        return (size_t)-1; /* 'invalid compression type' :121 */
    }
}

size_t zo_rawdeflate(const uint8_t *in, size_t n, int ctype, int lazy, uint8_t *out, size_t cap,
                     size_t out_index) {
    return rawdeflate_impl(in, n, ctype, lazy, 1, out, cap, out_index);
}
size_t zo_rawdeflate_heap32(const uint8_t *in, size_t n, int ctype, int lazy, uint8_t *out,
                            size_t cap, size_t out_index) {
    return rawdeflate_impl(in, n, ctype, lazy, 0, out, cap, out_index);
}

/* Many independent units in one call (bench.py's all-core baseline leg: one call per thread and
 * job keeps the interpreter out of the timing).  Returns the summed compressed size, or
 * (size_t)-1 on failure.  scratch must hold zo_deflate_bound(largest unit) bytes. */
size_t zo_rawdeflate_many(const uint8_t *in, const uint64_t *off, size_t n_units, int ctype, int lazy,
                          uint8_t *scratch, size_t scratch_cap) {
    size_t total = 0;
    for (size_t i = 0; i < n_units; ++i) {
        const size_t r = zo_rawdeflate(in + off[i], (size_t)(off[i + 1] - off[i]), ctype, lazy, scratch, scratch_cap, 0);
        if (r == (size_t)-1) return r;
        total += r;
    }
    return total;
}
