/*
 * zo_inflate.c -- CPU oracle: strict RFC 1951 / 1950 / 1952 decoders.
 * TEST INFRASTRUCTURE ONLY (see zo.h).
 *
 * The reference's own inflater (src/rawinflate.js:135-416, src/huffman.js)
 * is lenient: its stored-block LEN/NLEN test is vacuous (:233), it accepts
 * over-subscribed and incomplete code sets, and it needs trailing pad bytes
 * (:177).  The GPU box has no copy of the reference, so streams are verified
 * there by this decoder, which accepts strictly less than the reference
 * does: every stream it accepts is a conformant DEFLATE stream.  It decodes
 * the same block types in the same order (parseBlock :149-166; stored
 * :220-260; fixed :264-275; dynamic :279-330; symbol loop :380-416).
 */
#include "zo.h"
#include <string.h>

typedef struct {
    const uint8_t *in;
    size_t n, ip;
    uint64_t bitbuf;
    int bitcnt;
    uint8_t *out;
    size_t cap, op;
} zo_inf;

typedef struct {
    uint16_t count[16];
    uint16_t symbol[288];
} zo_huff;

static int need_bits(zo_inf *s, int need) {
    while (s->bitcnt < need) {
        if (s->ip >= s->n) return ZO_E_TRUNC;
        s->bitbuf |= (uint64_t)s->in[s->ip++] << s->bitcnt;
        s->bitcnt += 8;
    }
    return 0;
}
static int get_bits(zo_inf *s, int need, uint32_t *v) {
    if (need == 0) {
        *v = 0;
        return 0;
    }
    int e = need_bits(s, need);
    if (e) return e;
    *v = (uint32_t)(s->bitbuf & ((1ull << need) - 1));
    s->bitbuf >>= need;
    s->bitcnt -= need;
    return 0;
}

/* canonical decode, one bit at a time (counts per length, symbols in order) */
static int decode_sym(zo_inf *s, const zo_huff *h, int *sym) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        uint32_t b;
        int e = get_bits(s, 1, &b);
        if (e) return e;
        code |= (int)b;
        int count = h->count[len];
        if (code - count < first) {
            *sym = h->symbol[index + (code - first)];
            return 0;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return ZO_E_SYMBOL;
}

/* returns 0 for a complete set, >0 for incomplete, <0 over-subscribed */
static int build_huff(zo_huff *h, const uint8_t *length, int n) {
    uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[length[i]]++;
    if (h->count[0] == n) return 0; /* no codes at all: complete by convention, unusable */
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + h->count[len]);
    for (int i = 0; i < n; i++)
        if (length[i] != 0) h->symbol[offs[length[i]]++] = (uint16_t)i;
    return left;
}

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                   35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                   257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                   8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* token counts of the last decoded stream (diagnostics: tools/token_stats.py) */
static uint32_t g_tok_lits, g_tok_matches;
void zo_inflate_last_tokens(uint32_t out[2]) {
    out[0] = g_tok_lits;
    out[1] = g_tok_matches;
}

/* Per-block record of the dynamic blocks of the last zo_inflate_raw* call, when asked for with
 * zo_inflate_record_blocks: the code lengths the header carries and how often each symbol was used
 * (tests weigh a stream's code lengths against the reference's package-merge on the same counts). */
static zo_block_stats *g_blocks;
static int g_blocks_cap, g_blocks_n;
static zo_block_stats *g_cur_block;
void zo_inflate_record_blocks(zo_block_stats *blocks, int cap) {
    g_blocks = blocks;
    g_blocks_cap = cap;
    g_blocks_n = 0;
}
int zo_inflate_recorded_blocks(void) { return g_blocks_n; }

static int inflate_codes(zo_inf *s, const zo_huff *lc, const zo_huff *dc) {
    for (;;) {
        int sym;
        int e = decode_sym(s, lc, &sym);
        if (e) return e;
        if (g_cur_block) g_cur_block->litlen_freq[sym]++;
        if (sym < 256) {
            if (s->op >= s->cap) return ZO_E_OUTCAP;
            s->out[s->op++] = (uint8_t)sym;
            g_tok_lits++;
        } else if (sym == 256)
            return 0;
        else {
            sym -= 257;
            if (sym >= 29) return ZO_E_SYMBOL;
            uint32_t x;
            e = get_bits(s, LEXT[sym], &x);
            if (e) return e;
            size_t len = LBASE[sym] + x;
            g_tok_matches++;
            int ds;
            e = decode_sym(s, dc, &ds);
            if (e) return e;
            if (ds >= 30) return ZO_E_SYMBOL;
            if (g_cur_block) g_cur_block->dist_freq[ds]++;
            e = get_bits(s, DEXT[ds], &x);
            if (e) return e;
            size_t dist = DBASE[ds] + x;
            if (dist > s->op) return ZO_E_DIST;
            if (s->op + len > s->cap) return ZO_E_OUTCAP;
            for (size_t k = 0; k < len; k++, s->op++) s->out[s->op] = s->out[s->op - dist];
        }
    }
}

static int inflate_fixed(zo_inf *s) {
    static zo_huff lc, dc;
    static int ready;
    if (!ready) {
        uint8_t l[288];
        int i = 0;
        for (; i < 144; i++) l[i] = 8;
        for (; i < 256; i++) l[i] = 9;
        for (; i < 280; i++) l[i] = 7;
        for (; i < 288; i++) l[i] = 8;
        build_huff(&lc, l, 288);
        for (i = 0; i < 30; i++) l[i] = 5;
        build_huff(&dc, l, 30);
        ready = 1;
    }
    return inflate_codes(s, &lc, &dc);
}

static int inflate_dynamic(zo_inf *s) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
                                      11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t lengths[320];
    zo_huff lc, dc;
    uint32_t v;
    int e;
    if ((e = get_bits(s, 5, &v))) return e;
    int nlen = (int)v + 257;
    if ((e = get_bits(s, 5, &v))) return e;
    int ndist = (int)v + 1;
    if ((e = get_bits(s, 4, &v))) return e;
    int ncode = (int)v + 4;
    if (nlen > 286 || ndist > 30) return ZO_E_CODELENS;
    int idx = 0;
    for (; idx < ncode; idx++) {
        if ((e = get_bits(s, 3, &v))) return e;
        lengths[order[idx]] = (uint8_t)v;
    }
    for (; idx < 19; idx++) lengths[order[idx]] = 0;
    if (build_huff(&lc, lengths, 19) != 0) return ZO_E_CODELENS; /* must be complete */
    idx = 0;
    while (idx < nlen + ndist) {
        int sym;
        if ((e = decode_sym(s, &lc, &sym))) return e;
        if (sym < 16)
            lengths[idx++] = (uint8_t)sym;
        else {
            int len = 0, rep;
            if (sym == 16) {
                if (idx == 0) return ZO_E_CODELENS;
                len = lengths[idx - 1];
                if ((e = get_bits(s, 2, &v))) return e;
                rep = 3 + (int)v;
            } else if (sym == 17) {
                if ((e = get_bits(s, 3, &v))) return e;
                rep = 3 + (int)v;
            } else {
                if ((e = get_bits(s, 7, &v))) return e;
                rep = 11 + (int)v;
            }
            if (idx + rep > nlen + ndist) return ZO_E_CODELENS;
            while (rep--) lengths[idx++] = (uint8_t)len;
        }
    }
    if (lengths[256] == 0) return ZO_E_CODELENS;
    int err = build_huff(&lc, lengths, nlen);
    if (err < 0 || (err > 0 && nlen - lc.count[0] != 1)) return ZO_E_CODELENS;
    err = build_huff(&dc, lengths + nlen, ndist);
    if (err < 0 || (err > 0 && ndist - dc.count[0] != 1)) return ZO_E_CODELENS;
    g_cur_block = NULL;
    if (g_blocks && g_blocks_n < g_blocks_cap) {
        g_cur_block = &g_blocks[g_blocks_n++];
        memset(g_cur_block, 0, sizeof *g_cur_block);
        memcpy(g_cur_block->litlen_len, lengths, (size_t)nlen);
        memcpy(g_cur_block->dist_len, lengths + nlen, (size_t)ndist);
    }
    e = inflate_codes(s, &lc, &dc);
    g_cur_block = NULL;
    return e;
}

static int inflate_stored(zo_inf *s) {
    s->bitbuf = 0;
    s->bitcnt = 0;
    if (s->ip + 4 > s->n) return ZO_E_TRUNC;
    uint32_t len = s->in[s->ip] | ((uint32_t)s->in[s->ip + 1] << 8);
    uint32_t nlen = s->in[s->ip + 2] | ((uint32_t)s->in[s->ip + 3] << 8);
    s->ip += 4;
    if (len != (~nlen & 0xffff)) return ZO_E_NLEN;
    if (s->ip + len > s->n) return ZO_E_TRUNC;
    if (s->op + len > s->cap) return ZO_E_OUTCAP;
    memcpy(s->out + s->op, s->in + s->ip, len);
    s->ip += len;
    s->op += len;
    return 0;
}

int zo_inflate_raw_stats(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len,
                         size_t *in_used, uint32_t counts[3]) {
    zo_inf s;
    s.in = in;
    s.n = n;
    s.ip = 0;
    s.bitbuf = 0;
    s.bitcnt = 0;
    s.out = out;
    s.cap = cap;
    s.op = 0;
    if (counts) counts[0] = counts[1] = counts[2] = 0;
    g_tok_lits = g_tok_matches = 0;
    int last, e;
    do {
        uint32_t v;
        if ((e = get_bits(&s, 1, &v))) return e;
        last = (int)v;
        if ((e = get_bits(&s, 2, &v))) return e;
        if (v == 3) return ZO_E_BTYPE;
        if (counts) counts[v]++;
        e = v == 0 ? inflate_stored(&s) : v == 1 ? inflate_fixed(&s) : inflate_dynamic(&s);
        if (e) return e;
    } while (!last);
    /* whole bytes still buffered were not consumed */
    s.ip -= (size_t)(s.bitcnt / 8);
    if (out_len) *out_len = s.op;
    if (in_used) *in_used = s.ip;
    return 0;
}

int zo_inflate_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len,
                   size_t *in_used) {
    return zo_inflate_raw_stats(in, n, out, cap, out_len, in_used, NULL);
}

/* RFC 1950 wrapper; the reference's reader is src/inflate.js:17-58 */
int zo_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (n < 6) return ZO_E_TRUNC;
    if ((in[0] & 0x0f) != 8 || (in[0] >> 4) > 7) return ZO_E_HEADER;
    if (((in[0] << 8) | in[1]) % 31 != 0) return ZO_E_HEADER;
    if (in[1] & 0x20) return ZO_E_HEADER; /* FDICT unsupported */
    size_t ol = 0, used = 0;
    int e = zo_inflate_raw(in + 2, n - 2, out, cap, &ol, &used);
    if (e) return e;
    if (2 + used + 4 > n) return ZO_E_TRUNC;
    const uint8_t *t = in + 2 + used;
    uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    if (zo_adler32_update(1, out, ol) != want) return ZO_E_CHECKSUM;
    if (2 + used + 4 != n) return ZO_E_SIZE; /* trailing garbage */
    if (out_len) *out_len = ol;
    return 0;
}

/* RFC 1952 reader, all members; the reference's is src/gunzip.js:30-131 */
int zo_gunzip(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    size_t ip = 0, op = 0;
    while (ip < n) {
        if (n - ip < 18) return ZO_E_TRUNC;
        if (in[ip] != 0x1f || in[ip + 1] != 0x8b || in[ip + 2] != 8) return ZO_E_HEADER;
        int flg = in[ip + 3];
        size_t hdr0 = ip;
        ip += 10;
        if (flg & 0x04) {
            if (ip + 2 > n) return ZO_E_TRUNC;
            size_t xlen = in[ip] | ((size_t)in[ip + 1] << 8);
            ip += 2 + xlen;
        }
        if (flg & 0x08) {
            while (ip < n && in[ip]) ip++;
            ip++;
        }
        if (flg & 0x10) {
            while (ip < n && in[ip]) ip++;
            ip++;
        }
        if (flg & 0x02) {
            if (ip + 2 > n) return ZO_E_TRUNC;
            uint32_t want = in[ip] | ((uint32_t)in[ip + 1] << 8);
            if ((zo_crc32_update(in + hdr0, ip - hdr0, 0) & 0xffff) != want) return ZO_E_CHECKSUM;
            ip += 2;
        }
        if (ip >= n) return ZO_E_TRUNC;
        size_t ol = 0, used = 0;
        int e = zo_inflate_raw(in + ip, n - ip, out + op, cap - op, &ol, &used);
        if (e) return e;
        ip += used;
        if (ip + 8 > n) return ZO_E_TRUNC;
        uint32_t crc = in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16) |
                       ((uint32_t)in[ip + 3] << 24);
        uint32_t isize = in[ip + 4] | ((uint32_t)in[ip + 5] << 8) | ((uint32_t)in[ip + 6] << 16) |
                         ((uint32_t)in[ip + 7] << 24);
        ip += 8;
        if (zo_crc32_update(out + op, ol, 0) != crc) return ZO_E_CHECKSUM;
        if ((uint32_t)ol != isize) return ZO_E_SIZE;
        op += ol;
    }
    if (out_len) *out_len = op;
    return 0;
}
