/*
 * policy_sim.c -- CPU model of the search kernel's match selection and parse, used to
 * compare candidate-selection policies by compressed size before they are written as
 * GPU code.  Diagnostic tool only: it links the test oracle for the data generators and
 * the reference sizes and is never part of the product.
 *
 * Model (mirrors zlib.es_amd/csrc/deflate_kernel.hip): chains keyed by a 13-bit hash of
 * four bytes, linked in position order; items of 64 positions; pass 1 walks a bounded
 * number of candidates comparing up to CAP bytes; pass 2 scores shared distances for
 * the whole item with exact lengths; successor rule greedy / one-step lazy without
 * look-ahead across the item boundary; one block per 64 KiB.
 *
 * build: gcc -O2 -o /tmp/policy_sim tools/policy_sim.c -Ioracle -Loracle -lzoracle -Wl,-rpath,$PWD/oracle
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zo.h"

#define HBITS 13
#define MAXDIST (32768 - 64 * 54 - 128) /* the reach of the chain links; the kernel's one far candidate beyond it and its
                                           three-byte stride scoring are not modelled */
#define MAXN (1 << 21)

typedef struct {
    int chain;        /* candidates per walk */
    int cap;          /* pass-1 compare cap */
    int stop_at_cap;  /* 1: first capped candidate ends the walk (the shipped default) */
    int share_min;    /* offers need at least this length */
    int share_iters;  /* distinct distances scored per item */
    int share_order;  /* 0 lowest lane first, 1 highest lane first, 2 longest first */
    int keep;         /* capped candidates remembered per lane beyond the best (offers) */
    int deep;         /* extra candidates walked by "start" lanes with the tail check */
    int lazy;         /* parse: defer when next is longer and len < lazy */
    int min3dist;     /* drop length-3 matches farther than this */
    int dist1;        /* also score distance 1 for the item */
    int exact_own;    /* every lane's own capped best gets its exact length */
    int start_only;   /* deep walk only on start lanes */
    int tail_all;     /* 1: after the first capped candidate, go on walking `chain` candidates with the tail check */
    int fan;          /* probe-ahead fans per item: the distances of the nearest earlier occurrences of the keys at p + j, tried at p */
    int fan_min;      /* a lane fans when its best is at least this long and does not continue its predecessor's match */
    int fan_stride;   /* j advances by this */
    int fan_try;      /* passing candidates verified per fan, farthest j first (0 = all) */
    int fan_from0;    /* 1: probes start at j = 0 instead of at the tail of the current best */
    int fan_maxd;     /* a lane fans only when its best distance is at most this (0 = any) */
    int fan_depth;    /* links followed along the chain of each probed position (0 = 1) */
    int cert;         /* 1: a deep walk ends (or does not start) when the four bytes that straddle the end of the best match have no earlier occurrence: nothing longer exists */
    int run_stop;     /* > 0: after this many candidates the walk ends for the whole item when at least run_frac lanes hold a capped match */
    int run_frac;
    int lane0;        /* 1: the item's first position is not a start lane when the byte before it repeats at its best distance */
    int run_same;     /* > 0: an item with at least this many such lanes walks run_stop candidates */
    int tailwalk;     /* > 0: start lanes walk this many candidates along the chain of the straddle position */
    int tail_lane0;   /* 1: the item's first position tail-walks even when it continues a match */
    int tail_maxd;    /* tail walks only from matches at most this far back (0 = any) */
    int seqrule;      /* 1: a deep walk ends when its chain runs through four consecutive positions */
    int maxstart;     /* 1: a position whose predecessor's match has the maximum length is a start lane for the tail walk */
    int tail_restart; /* after an improvement: 0 go on along the same chain, 1 start over at the new straddle position, 2 go on, and start over when the chain ends */
    int run_lag;      /* > 0: the rule looks at the item this many items back instead of at the item itself */
    int itemfast;     /* 1: an item whose 64 positions all repeat at lane 0's nearest candidate distance takes that distance and skips the search */
} Policy;
static uint64_t g_fans, g_fast;

static uint8_t data[MAXN + 512];
static int32_t prevp[MAXN];
static int32_t headp[1 << HBITS];
static uint32_t best_len[MAXN], best_dist[MAXN];
static uint8_t item_flag[MAXN / 64 + 1];

static inline uint32_t hash4(const uint8_t *p) {
    uint32_t w;
    memcpy(&w, p, 4);
    return (w * 0x9E3779B1u) >> (32 - HBITS);
}
static inline uint32_t mlen(const uint8_t *a, const uint8_t *b, uint32_t max) {
    uint32_t l = 0;
    while (l < max && a[l] == b[l]) ++l;
    return l;
}

/* ---- Huffman cost (lengths limited by simple depth clamp + Kraft repair, like the kernel) */
static void huff_lengths(const uint32_t *f, int n, int limit, uint8_t *len) {
    int idx[300], m = 0;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (f[i]) idx[m++] = i;
    }
    if (m == 0) return;
    if (m == 1) {
        len[idx[0]] = 1;
        return;
    }
    /* sort ascending by freq */
    for (int i = 1; i < m; ++i) {
        int v = idx[i], j = i;
        while (j > 0 && (f[idx[j - 1]] > f[v] || (f[idx[j - 1]] == f[v] && idx[j - 1] > v))) {
            idx[j] = idx[j - 1];
            --j;
        }
        idx[j] = v;
    }
    uint64_t w[600];
    int par[600];
    for (int i = 0; i < m; ++i) w[i] = f[idx[i]];
    int li = 0, ni = m, nn = m;
    for (int k = 0; k + 1 < m; ++k) {
        int a, b;
        if (li < m && (ni >= nn || w[li] <= w[ni])) a = li++; else a = ni++;
        if (li < m && (ni >= nn || w[li] <= w[ni])) b = li++; else b = ni++;
        w[nn] = w[a] + w[b];
        par[a] = nn;
        par[b] = nn;
        ++nn;
    }
    int dep[600];
    dep[nn - 1] = 0;
    for (int i = nn - 2; i >= 0; --i) dep[i] = dep[par[i]] + 1;
    uint32_t blc[64] = {0};
    int over = 0;
    for (int i = 0; i < m; ++i) {
        int d = dep[i];
        if (d > limit) {
            d = limit;
            over = 1;
        }
        blc[d]++;
    }
    if (over) {
        uint32_t kraft = 0;
        for (int b = 1; b <= limit; ++b) kraft += blc[b] << (limit - b);
        for (uint32_t ex = kraft - (1u << limit); ex > 0; --ex) {
            int bits = limit - 1;
            while (blc[bits] == 0) --bits;
            blc[bits]--;
            blc[bits + 1] += 2;
            blc[limit]--;
        }
    }
    /* rarest symbols get the longest codes */
    int i = 0;
    for (int b = limit; b >= 1; --b)
        for (uint32_t c = 0; c < blc[b]; ++c) len[idx[i++]] = (uint8_t)b;
}

static const int LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const int LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static int lsym_of[259], dsym_of_small[512];
static int dist_sym(uint32_t d, int *eb) {
    uint32_t dm = d - 1;
    if (dm < 4) {
        *eb = 0;
        return (int)dm;
    }
    int nbit = 31 - __builtin_clz(dm);
    *eb = nbit - 1;
    return 2 * nbit + ((dm >> (nbit - 1)) & 1);
}

typedef struct {
    uint32_t fl[288], fd[32];
    uint64_t extra_bits;
    uint64_t ntok;
} Blk;

static uint64_t header_bits(const uint8_t *ll, const uint8_t *dl) {
    int hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) --hlit;
    while (hdist > 1 && dl[hdist - 1] == 0) --hdist;
    uint8_t seq[320];
    int n = 0;
    for (int i = 0; i < hlit; ++i) seq[n++] = ll[i];
    for (int i = 0; i < hdist; ++i) seq[n++] = dl[i];
    uint32_t cf[19] = {0};
    uint64_t extra = 0;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && seq[j] == seq[i]) ++j;
        int r = j - i;
        if (seq[i] == 0) {
            while (r >= 11) {
                int t = r > 138 ? 138 : r;
                cf[18]++;
                extra += 7;
                r -= t;
            }
            if (r >= 3) {
                cf[17]++;
                extra += 3;
                r = 0;
            }
            cf[0] += r;
        } else {
            cf[seq[i]]++;
            --r;
            while (r >= 3) {
                int t = r > 6 ? 6 : r;
                cf[16]++;
                extra += 2;
                r -= t;
            }
            cf[seq[i]] += r;
        }
        i = j;
    }
    uint8_t cl[19];
    int nz = 0;
    for (int i = 0; i < 19; ++i) nz += cf[i] != 0;
    if (nz < 2) cf[cf[0] ? 1 : 0] = 1;
    huff_lengths(cf, 19, 7, cl);
    static const int ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl[ord[hclen - 1]] == 0) --hclen;
    uint64_t bits = 17 + 3 * hclen + extra;
    for (int i = 0; i < 19; ++i) bits += (uint64_t)cf[i] * cl[i];
    return bits;
}

static uint64_t block_cost(Blk *b, int ctype) {
    b->fl[256] = 1;
    if (ctype == 1) {
        uint64_t bits = 3;
        for (int s = 0; s < 288; ++s) bits += (uint64_t)b->fl[s] * (s <= 143 ? 8 : s <= 255 ? 9 : s <= 279 ? 7 : 8);
        for (int s = 0; s < 30; ++s) bits += (uint64_t)b->fd[s] * 5;
        return bits + b->extra_bits;
    }
    uint8_t ll[288], dl[32];
    uint32_t fl[288], fd[32];
    memcpy(fl, b->fl, sizeof fl);
    memcpy(fd, b->fd, sizeof fd);
    if (b->ntok < 1) fl[0] = 1;
    int nzd = 0;
    for (int s = 0; s < 30; ++s) nzd += fd[s] != 0;
    if (nzd == 0) fd[0] = fd[1] = 1;
    else if (nzd == 1) fd[fd[0] ? 1 : 0] = 1;
    huff_lengths(fl, 286, 15, ll);
    huff_lengths(fd, 30, 15, dl);
    uint64_t bits = header_bits(ll, dl);
    for (int s = 0; s < 286; ++s) bits += (uint64_t)b->fl[s] * ll[s];
    for (int s = 0; s < 30; ++s) bits += (uint64_t)b->fd[s] * dl[s];
    return bits + b->extra_bits;
}

static int g_dump = 0;
static uint32_t g_pos = 0;
static void tok_lit(Blk *b, uint8_t c) {
    if (g_dump) printf("(%u L %u) ", g_pos, c);
    b->fl[c]++;
    b->ntok++;
}
static void tok_match(Blk *b, uint32_t len, uint32_t dist) {
    if (g_dump) printf("(%u M %u %u) ", g_pos, len, dist);
    int ls = lsym_of[len];
    b->fl[257 + ls]++;
    b->extra_bits += LEXT[ls];
    int eb, ds = dist_sym(dist, &eb);
    b->fd[ds]++;
    b->extra_bits += eb;
    b->ntok++;
}

/* exact match length of position p at distance d (0 if d is out of reach) */
static uint32_t len_at(uint32_t p, uint32_t d, uint32_t n) {
    uint32_t limit = p < MAXDIST ? p : MAXDIST;
    if (d == 0 || d > limit) return 0;
    uint32_t maxlen = n - p < 258 ? n - p : 258;
    return mlen(data + p, data + p - d, maxlen);
}

static uint64_t simulate(uint32_t n, const Policy *P, int ctype, uint64_t *walk_steps, uint64_t *share_steps) {
    for (int i = 0; i < (1 << HBITS); ++i) headp[i] = -1;
    for (uint32_t p = 0; p < n; ++p) {
        prevp[p] = -1;
        if (p + 4 <= n) {
            uint32_t h = hash4(data + p);
            prevp[p] = headp[h];
            headp[h] = (int32_t)p;
        }
    }
    uint64_t steps = 0, shares = 0;
    for (uint32_t p0 = 0; p0 < n; p0 += 64) {
        uint32_t cnt = n - p0 < 64 ? n - p0 : 64;
        uint32_t blen[64], bd[64], offers[64][8], noff[64];
        int32_t cursor[64];
        int budget_left[64];
        for (uint32_t l = 0; l < 64; ++l) {
            blen[l] = 2;
            bd[l] = 0;
            noff[l] = 0;
            cursor[l] = -1;
            budget_left[l] = 0;
        }
        int maxsteps = 0;
        if (P->itemfast && cnt == 64 && p0 + 4 <= n && prevp[p0] >= 0) {
            uint32_t d0 = p0 - (uint32_t)prevp[p0];
            uint32_t lim0 = p0 < MAXDIST ? p0 : MAXDIST;
            int all = d0 <= lim0;
            for (uint32_t l = 0; l < 64 && all; ++l) all = data[p0 + l] == data[p0 + l - d0];
            if (all) {
                for (uint32_t l = 0; l < 64; ++l) {
                    best_len[p0 + l] = len_at(p0 + l, d0, n);
                    best_dist[p0 + l] = d0;
                    if (best_len[p0 + l] < 3) best_len[p0 + l] = 0;
                }
                ++g_fast;
                steps += 1;
                continue;
            }
        }
        /* the lanes of an item walk in lock-step (as the wave does), so that rules that look at the
         * whole item between candidates can be modelled */
        {
            int32_t wc[64];
            uint32_t wlast[64];
            int walking[64], capseen[64], wst[64], wbudget[64];
            for (uint32_t l = 0; l < 64; ++l) {
                walking[l] = 0;
                capseen[l] = 0;
                wst[l] = 0;
                wbudget[l] = P->chain;
                wlast[l] = 0;
                wc[l] = -1;
            }
            for (uint32_t l = 0; l < cnt; ++l) {
                if (p0 + l + 4 > n) continue;
                walking[l] = 1;
                wc[l] = prevp[p0 + l];
            }
            int chain_here = P->chain;
            if (P->run_lag && P->run_stop) {
                /* the item run_lag items back (parsed before this one may start) was run-like: this one
                 * walks run_stop candidates only */
                uint32_t kk = p0 >> 6;
                if (kk >= (uint32_t)P->run_lag && item_flag[kk - P->run_lag]) chain_here = P->run_stop;
            }
            if (P->run_same && P->run_stop) {
                /* lanes whose nearest candidate lies at the same distance as their left neighbour's: inside copies */
                int same = 0;
                for (uint32_t l = 1; l < cnt; ++l)
                    if (walking[l] && walking[l - 1] && wc[l] >= 0 && wc[l - 1] >= 0 && wc[l] == wc[l - 1] + 1) ++same;
                if (same >= P->run_same) chain_here = P->run_stop;
            }
            for (int step = 0; step < chain_here; ++step) {
                if (!P->run_lag && P->run_stop && step == P->run_stop) {
                    /* run-like item: at least run_frac lanes hold a capped match -- the others lie in
                     * the shadow of long matches and stop walking */
                    int capped = 0;
                    for (uint32_t l = 0; l < cnt; ++l) capped += blen[l] >= (uint32_t)P->cap;
                    if (capped >= P->run_frac) break;
                }
                int any = 0;
                for (uint32_t l = 0; l < cnt; ++l) {
                    if (!walking[l]) continue;
                    uint32_t p = p0 + l;
                    uint32_t maxlen = n - p < 258 ? n - p : 258;
                    uint32_t limit = p < MAXDIST ? p : MAXDIST;
                    uint32_t cmpmax = maxlen < (uint32_t)P->cap ? maxlen : (uint32_t)P->cap;
                    int32_t c = wc[l];
                    if (c < 0) { walking[l] = 0; continue; }
                    uint32_t dist = p - (uint32_t)c;
                    if (dist <= wlast[l] || dist > limit) { walking[l] = 0; continue; }
                    wlast[l] = dist;
                    ++wst[l];
                    any = 1;
                    uint32_t len;
                    if (capseen[l] && P->tail_all) {
                        uint32_t bl = blen[l];
                        if (bl < maxlen && data[c + bl] == data[p + bl] && data[c + bl - 1] == data[p + bl - 1])
                            len = mlen(data + p, data + c, maxlen);
                        else
                            len = 0;
                    } else {
                        len = mlen(data + p, data + c, cmpmax);
                    }
                    int32_t nxt = prevp[c];
                    --wbudget[l];
                    if (len > blen[l]) {
                        blen[l] = len;
                        bd[l] = dist;
                        if (len >= maxlen) {
                            wc[l] = nxt;
                            walking[l] = 0;
                            continue;
                        }
                        if (len >= cmpmax && !capseen[l]) {
                            capseen[l] = 1;
                            if (P->stop_at_cap) {
                                wc[l] = nxt;
                                walking[l] = 0;
                                continue;
                            }
                            if (P->tail_all) blen[l] = mlen(data + p, data + c, maxlen);
                        }
                    } else if (len >= cmpmax && capseen[l] && !P->tail_all && noff[l] < (uint32_t)P->keep) {
                        offers[l][noff[l]++] = dist;
                    }
                    wc[l] = nxt;
                }
                if (any) ++maxsteps;
            }
            {
                int capped = 0;
                for (uint32_t l = 0; l < cnt; ++l) capped += blen[l] >= (uint32_t)P->cap;
                item_flag[p0 >> 6] = capped >= P->run_frac;
            }
            for (uint32_t l = 0; l < cnt; ++l) {
                cursor[l] = wc[l];
                budget_left[l] = wbudget[l];
            }
        }
        steps += (uint64_t)maxsteps;
        /* pass 2: share distances */
        {
            uint32_t cand[64 * 9], ncand = 0;
            uint32_t candlen[64 * 9];
            if (P->share_order == 1) {
                for (int l = (int)cnt - 1; l >= 0; --l) {
                    if (blen[l] >= (uint32_t)P->share_min) {
                        candlen[ncand] = blen[l];
                        cand[ncand++] = bd[l];
                    }
                    for (uint32_t k = 0; k < noff[l]; ++k) {
                        candlen[ncand] = P->cap;
                        cand[ncand++] = offers[l][k];
                    }
                }
            } else {
                for (uint32_t l = 0; l < cnt; ++l) {
                    if (blen[l] >= (uint32_t)P->share_min) {
                        candlen[ncand] = blen[l];
                        cand[ncand++] = bd[l];
                    }
                    for (uint32_t k = 0; k < noff[l]; ++k) {
                        candlen[ncand] = P->cap;
                        cand[ncand++] = offers[l][k];
                    }
                }
            }
            if (P->dist1) {
                candlen[ncand] = 0;
                cand[ncand++] = 1;
            }
            uint32_t done[64];
            int nd = 0;
            for (uint32_t i = 0; i < ncand && nd < P->share_iters; ++i) {
                uint32_t d = cand[i];
                int dup = 0;
                for (int k = 0; k < nd; ++k) dup |= done[k] == d;
                if (dup) continue;
                done[nd++] = d;
                ++shares;
                for (uint32_t l = 0; l < cnt; ++l) {
                    uint32_t run = len_at(p0 + l, d, n);
                    if (run >= 3 && (run > blen[l] || (run == blen[l] && d < bd[l]))) {
                        blen[l] = run;
                        bd[l] = d;
                    }
                }
            }
            if (P->exact_own) {
                for (uint32_t l = 0; l < cnt; ++l)
                    if (blen[l] >= (uint32_t)P->cap) {
                        uint32_t run = len_at(p0 + l, bd[l], n);
                        if (run > blen[l]) blen[l] = run;
                    }
            }
        }
        /* pass 3: deep walk with the tail check */
        if (P->deep > 0 || P->tailwalk > 0) {
            int mx = 0;
            uint32_t newd[64];
            int nnew = 0;
            for (uint32_t l = 0; l < cnt; ++l) {
                uint32_t p = p0 + l;
                if (blen[l] < (uint32_t)P->cap) continue;
                if (P->start_only && l > 0 && bd[l - 1] == bd[l] && blen[l - 1] == blen[l] + 1) continue;
                /* the item's first position has no predecessor in the item: it continues a match from before when
                 * the byte in front of it repeats at the same distance */
                if (P->lane0 && l == 0 && p > bd[0] && data[p - 1] == data[p - 1 - bd[0]]) continue;
                uint32_t maxlen = n - p < 258 ? n - p : 258;
                uint32_t limit = p < MAXDIST ? p : MAXDIST;
                if (blen[l] >= maxlen) continue;
                int32_t c = cursor[l];
                uint32_t last = c >= 0 ? 0 : 0;
                int st = 0;
                uint32_t lastd = 0;
                int need_cert = 1;
                int seqn = 0;
                for (int b = P->deep; b > 0 && c >= 0; --b) {
                    if (P->seqrule && seqn >= 4) break;
                    if (P->cert && need_cert) {
                        need_cert = 0;
                        ++st;
                        uint32_t t = p + blen[l] - 3;
                        int32_t c2 = prevp[t];
                        if (c2 < 0 || t - (uint32_t)c2 > limit || memcmp(data + c2, data + t, 4)) break;
                    }
                    uint32_t dist = p - (uint32_t)c;
                    if (dist <= lastd || dist > limit) break;
                    seqn = dist == lastd + 1 ? seqn + 1 : 0;
                    lastd = dist;
                    ++st;
                    uint32_t bl = blen[l];
                    if (data[c + bl] == data[p + bl] && data[c + bl - 1] == data[p + bl - 1] &&
                        data[c + bl - 2] == data[p + bl - 2] && data[c + bl - 3] == data[p + bl - 3]) {
                        uint32_t len = mlen(data + p, data + c, maxlen);
                        if (len > bl) {
                            blen[l] = len;
                            bd[l] = dist;
                            need_cert = 1;
                            if (nnew < 64) newd[nnew++] = dist;
                            if (len >= maxlen) break;
                        }
                    }
                    c = prevp[c];
                }
                (void)last;
                if (st > mx) mx = st;
            }
            /* tail walk: along the chain of the position t where the four bytes that straddle the end of the best
             * match start; a candidate there lines up at distance d = t - c, and is measured when the match's head
             * agrees at that distance */
            if (P->tailwalk > 0) {
                int mx2 = 0;
                for (uint32_t l = 0; l < cnt; ++l) {
                    uint32_t p = p0 + l;
                    if (blen[l] < (uint32_t)P->cap) continue;
                    if (l > 0 && bd[l - 1] == bd[l] && blen[l - 1] == blen[l] + 1 && !(P->maxstart && blen[l - 1] >= 258)) continue;
                    if (P->tail_lane0 == 0 && l == 0 && p > bd[0] && data[p - 1] == data[p - 1 - bd[0]]) continue;
                    if (P->tail_maxd && bd[l] > (uint32_t)P->tail_maxd) continue;
                    uint32_t maxlen = n - p < 258 ? n - p : 258;
                    uint32_t limit = p < MAXDIST ? p : MAXDIST;
                    if (blen[l] >= maxlen) continue;
                    uint32_t t = p + blen[l] - 3;
                    if (t + 4 > n) continue;
                    int32_t c = prevp[t];
                    uint32_t lastd = 0, t_len = blen[l];
                    int st = 0;
                    for (int b = P->tailwalk; b > 0; --b) {
                        uint32_t d = c >= 0 ? t - (uint32_t)c : 0;
                        if (c < 0 || d <= lastd || d > limit) {
                            /* this chain is through: the one of the straddle position of the match as it is now */
                            if (P->tail_restart == 2 && blen[l] > t_len) {
                                t = p + blen[l] - 3;
                                t_len = blen[l];
                                if (t + 4 > n) break;
                                c = prevp[t];
                                lastd = 0;
                                continue;
                            }
                            break;
                        }
                        lastd = d;
                        ++st;
                        if (!memcmp(data + c, data + t, 4) && !memcmp(data + p - d, data + p, 4) &&
                            data[p + blen[l] - d] == data[p + blen[l]]) {
                            uint32_t len = mlen(data + p, data + p - d, maxlen);
                            if (len > blen[l]) {
                                blen[l] = len;
                                bd[l] = d;
                                if (nnew < 64) newd[nnew++] = d;
                                if (len >= maxlen) break;
                                if (P->tail_restart == 1) {
                                    t = p + len - 3;
                                    t_len = len;
                                    if (t + 4 > n) break;
                                    c = prevp[t];
                                    lastd = 0;
                                    continue;
                                }
                            }
                        }
                        c = prevp[c];
                    }
                    if (st > mx2) mx2 = st;
                }
                steps += (uint64_t)mx2;
            }
            steps += (uint64_t)mx;
            /* share what the deep walks found (later lanes of the same structure) */
            uint32_t done[64];
            int nd = 0;
            for (int i = nnew - 1; i >= 0 && nd < 4; --i) {
                uint32_t d = newd[i];
                int dup = 0;
                for (int k = 0; k < nd; ++k) dup |= done[k] == d;
                if (dup) continue;
                done[nd++] = d;
                ++shares;
                for (uint32_t l = 0; l < cnt; ++l) {
                    uint32_t run = len_at(p0 + l, d, n);
                    if (run >= 3 && (run > blen[l] || (run == blen[l] && d < bd[l]))) {
                        blen[l] = run;
                        bd[l] = d;
                    }
                }
            }
        }
        /* pass 3b: probe-ahead fan */
        {
            uint8_t fanned[64] = {0};
            for (int f = 0; f < P->fan; ++f) {
                int l = -1;
                for (uint32_t q = 0; q < cnt; ++q) {
                    uint32_t p = p0 + q, maxlen = n - p < 258 ? n - p : 258;
                    if (fanned[q] || blen[q] < (uint32_t)P->fan_min || blen[q] >= maxlen) continue;
                    if (P->fan_maxd && bd[q] > (uint32_t)P->fan_maxd) continue;
                    if (q > 0 && bd[q - 1] == bd[q] && blen[q - 1] == blen[q] + 1) continue;
                    l = (int)q;
                    break;
                }
                if (l < 0) break;
                fanned[l] = 1;
                ++g_fans;
                uint32_t p = p0 + l, maxlen = n - p < 258 ? n - p : 258, limit = p < MAXDIST ? p : MAXDIST;
                uint32_t pass_d[64];
                int npass = 0;
                uint32_t j0 = P->fan_from0 ? 0 : blen[l] - 3;
                for (int i = 0; i < 64; ++i) {
                    uint32_t j = j0 + (uint32_t)P->fan_stride * i;
                    if (j + 4 > maxlen) break;
                    int32_t c2 = prevp[p + j];
                    for (int dep = 0; dep < (P->fan_depth ? P->fan_depth : 1) && c2 >= 0; ++dep, c2 = prevp[c2]) {
                        uint32_t d = p + j - (uint32_t)c2;
                        if (d > limit) break;
                        if (memcmp(data + p - d, data + p, 4) || memcmp(data + c2, data + p + j, 4)) continue;
                        if (npass < 64) pass_d[npass++] = d;
                        break; /* the nearest one of this chain that lines up */
                    }
                }
                uint32_t newd = 0;
                int tries = 0;
                for (int i = npass - 1; i >= 0; --i) {
                    if (P->fan_try && tries >= P->fan_try) break;
                    int dup = 0;
                    for (int k = i + 1; k < npass; ++k) dup |= pass_d[k] == pass_d[i];
                    if (dup) continue;
                    ++tries;
                    uint32_t len = len_at(p, pass_d[i], n);
                    if (len > blen[l]) {
                        blen[l] = len;
                        bd[l] = pass_d[i];
                        newd = pass_d[i];
                    }
                }
                if (newd) {
                    ++shares;
                    for (uint32_t q = 0; q < cnt; ++q) {
                        uint32_t run = len_at(p0 + q, newd, n);
                        if (run >= 3 && (run > blen[q] || (run == blen[q] && newd < bd[q]))) {
                            blen[q] = run;
                            bd[q] = newd;
                        }
                    }
                }
            }
        }
        for (uint32_t l = 0; l < cnt; ++l) {
            uint32_t bl = blen[l];
            if (bl < 3 || (bl == 3 && bd[l] > (uint32_t)P->min3dist)) bl = 0;
            best_len[p0 + l] = bl;
            best_dist[p0 + l] = bd[l];
        }
    }
    if (walk_steps) *walk_steps = steps;
    if (share_steps) *share_steps = shares;
    /* parse */
    uint64_t bits = 0;
    Blk blk;
    memset(&blk, 0, sizeof blk);
    uint32_t p = 0, blk_end = 65536;
    while (p < n) {
        if (p >= blk_end) {  /* blocks are cut at item granularity by entry position; close enough */
            bits += block_cost(&blk, ctype);
            memset(&blk, 0, sizeof blk);
            blk_end += 65536;
        }
        uint32_t bl = best_len[p];
        g_pos = p;
        uint32_t lane = p & 63;
        uint32_t l1 = (lane < 63 && p + 1 < n) ? best_len[p + 1] : 0;
        if (bl >= 3) {
            if (P->lazy > 0 && bl < (uint32_t)P->lazy && l1 > bl) {
                tok_lit(&blk, data[p]);
                tok_match(&blk, l1, best_dist[p + 1]);
                p += 1 + l1;
            } else {
                tok_match(&blk, bl, best_dist[p]);
                p += bl;
            }
        } else {
            tok_lit(&blk, data[p]);
            p += 1;
        }
    }
    bits += block_cost(&blk, ctype);
    return (bits + 7) / 8;
}

typedef struct {
    const char *name;
    int gen;  /* 0 rnd 1 seq 2 rndseq 3 text 4 runs 5 small-alpha */
    uint32_t n, seed, k;
} Case;

int main(int argc, char **argv) {
    for (int l = 3; l <= 258; ++l) {
        int s = 28;
        while (LBASE[s] > l) --s;
        if (l == 258) s = 28;
        lsym_of[l] = s;
    }
    (void)dsym_of_small;
    static const Case cases[] = {
        {"seq76543", 1, 76543, 0, 0},         {"rndseq76543", 2, 76543, (uint32_t)1346432776267ull, 0},
        {"rndseq65536", 2, 65536, 1, 0},      {"seq123456", 1, 123456, 0, 0},
        {"text65536", 3, 65536, 42, 0},       {"text16384", 3, 16384, 7, 0},
        {"text262144", 3, 262144, 42, 0},     {"runs65536", 4, 65536, 3, 0},
        {"alpha2_20000", 5, 20000, 5, 2},     {"alpha3_70000", 5, 70000, 6, 3},
        {"rnd76543", 0, 76543, 1, 0},
    };
    const int ncases = sizeof cases / sizeof cases[0];
    /* policies: name=value pairs on the command line override the defaults below */
    Policy P;
    memset(&P, 0, sizeof P);
    P.chain = 8; P.cap = 32; P.stop_at_cap = 1; P.share_min = 16; P.share_iters = 6; P.lazy = 258; P.min3dist = 4096;
    P.start_only = 1; P.fan_min = 32; P.fan_stride = 4; P.run_frac = 32; P.seqrule = 1;
    for (int a = 1; a < argc; ++a) {
        char key[32];
        int v;
        if (sscanf(argv[a], "%31[^=]=%d", key, &v) != 2) continue;
#define K(name) if (!strcmp(key, #name)) P.name = v
        K(chain); K(cap); K(stop_at_cap); K(share_min); K(share_iters); K(share_order); K(keep); K(deep);
        K(lazy); K(min3dist); K(dist1); K(exact_own); K(start_only); K(tail_all);
        K(fan); K(fan_min); K(fan_stride); K(fan_try); K(fan_from0); K(fan_maxd); K(fan_depth); K(itemfast); K(cert); K(run_stop); K(run_frac); K(run_lag); K(tail_restart); K(maxstart); K(tailwalk); K(tail_lane0); K(tail_maxd); K(seqrule); K(run_same); K(lane0);
#undef K
    }
    g_dump = getenv("DUMP") != NULL;
    const char *only = getenv("ONLY");
    static uint8_t out[MAXN + MAXN / 4 + 1024];
    double tot[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    printf("%-14s %8s %8s %6s | %8s %8s %6s | walk/item share/item\n", "case", "refFX", "simFX", "ratio", "refDY", "simDY", "ratio");
    for (int i = 0; i < ncases; ++i) {
        const Case *c = &cases[i];
        if (only && strcmp(only, c->name)) continue;
        uint32_t n = c->n;
        uint32_t seed = getenv("SEED") ? (uint32_t)strtoul(getenv("SEED"), NULL, 10) : c->seed;
        switch (c->gen) {
        case 0: zo_gen_random(data, n, seed); break;
        case 1: zo_gen_sequential(data, n); break;
        case 2: zo_gen_random_sequential(data, n, seed); break;
        case 3: zo_gen_text(data, n, seed); break;
        case 4: zo_gen_runs(data, n, seed); break;
        case 5: zo_gen_small_alpha(data, n, seed, c->k); break;
        }
        memset(data + n, 0, 512);
        size_t r1 = zo_rawdeflate(data, n, 1, 0, out, sizeof out, 0);
        size_t r2 = zo_rawdeflate(data, n, 2, 0, out, sizeof out, 0);
        uint64_t ws = 0, ss = 0;
        g_fans = 0;
        g_fast = 0;
        uint64_t s1 = simulate(n, &P, 1, &ws, &ss);
        uint64_t fans = g_fans, fast = g_fast;
        uint64_t s2 = simulate(n, &P, 2, NULL, NULL);
        double items = (n + 63) / 64;
        printf("%-14s %8zu %8llu %6.3f | %8zu %8llu %6.3f | %5.2f %5.2f fans %4.2f fast %4.2f\n", c->name, r1, (unsigned long long)s1,
               (double)s1 / r1, r2, (unsigned long long)s2, (double)s2 / r2, ws / items, ss / items, fans / items, fast / items);
        tot[1][0] += r1;
        tot[1][1] += s1;
        tot[2][0] += r2;
        tot[2][1] += s2;
    }
    printf("TOTAL FX %.4f DY %.4f\n", tot[1][1] / tot[1][0], tot[2][1] / tot[2][0]);
    return 0;
}
