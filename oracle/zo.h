/*
 * zo.h -- CPU oracle for the zlib.es compression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain-C restatement of the
 * reference's JavaScript algorithms (file:line cited at each function) and a
 * strict RFC 1951/1950/1952 decoder used as the verifier.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it; the
 * product (libzgpu.so and the host shims) never links or loads it.
 *
 * Parity status: PINNED.  Every encoder function here is checked byte for
 * byte against outputs of the reference itself (src/rawdeflate.js,
 * src/crc32.js, src/adler32.js, src/deflate.js, src/gzip.js, src/zip.js run
 * under Node in the build container), committed under tests/golden/ by
 * tests/golden/gen_golden.mjs.
 */
#ifndef ZO_H
#define ZO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ZO_NONE = 0, ZO_FIXED = 1, ZO_DYNAMIC = 2 };

/* ---- checksums: src/crc32.js:14-50, src/adler32.js:11-37 ---- */
uint32_t zo_crc32_update(const uint8_t *data, size_t len, uint32_t crc);
uint32_t zo_crc32_single(uint32_t num, uint32_t crc);
uint32_t zo_adler32_update(uint32_t adler, const uint8_t *data, size_t len);

/* ---- RawDeflate: src/rawdeflate.js ---- */
/* LZ77 token stream exactly as RawDeflate.lz77 builds it (Uint16 words;
 * literal = 1 word, match = 6 words, EOB = 256).  tokens must hold 2*n+8
 * words.  Returns the word count.  freqs may be NULL. */
size_t zo_lz77(const uint8_t *in, size_t n, int lazy, uint16_t *tokens,
               uint32_t *freqs_litlen /*286*/, uint32_t *freqs_dist /*30*/);
/* getLengths_ (Heap + reversePackageMerge_).  heap16!=0 reproduces the
 * Uint16Array heap storage of src/heap.js:11 (the reference behaviour). */
void zo_get_lengths(const uint32_t *freqs, int nsym, int limit, int heap16,
                    uint8_t *lengths);
/* getCodesFromLengths_: canonical codes, bit-mirrored. */
void zo_get_codes(const uint8_t *lengths, int nsym, uint16_t *codes);
/* RawDeflate.compress().  The first out_index bytes of out are left as the
 * caller's header (outputBuffer/outputIndex semantics).  Returns total bytes
 * (header included) or (size_t)-1 when cap is too small. */
size_t zo_rawdeflate(const uint8_t *in, size_t n, int ctype, int lazy,
                     uint8_t *out, size_t cap, size_t out_index);
/* Same with a 32-bit heap (the reference's Uint16 wrap removed); used only to
 * quantify the quirk, never as the parity target. */
size_t zo_rawdeflate_heap32(const uint8_t *in, size_t n, int ctype, int lazy,
                            uint8_t *out, size_t cap, size_t out_index);
size_t zo_deflate_bound(size_t n);
/* n_units independent RawDeflate.compress() calls; returns the summed size */
size_t zo_rawdeflate_many(const uint8_t *in, const uint64_t *off, size_t n_units, int ctype, int lazy,
                          uint8_t *scratch, size_t scratch_cap);

/* ---- containers: src/deflate.js:44-99, src/gzip.js:60-131 ---- */
size_t zo_zlib_compress(const uint8_t *in, size_t n, int ctype, int lazy,
                        uint8_t *out, size_t cap);
/* flags: bit0 fname, bit1 fcomment, bit2 fhcrc (the reference's option
 * gates, see zo_containers.c). */
size_t zo_gzip_compress(const uint8_t *in, size_t n, int ctype, int lazy,
                        uint32_t mtime, int flags, const char *filename,
                        const char *comment, uint8_t *out, size_t cap);

/* ---- strict decoders (verifier; design of this repo, RFC 1951/1950/1952) --
 * Return 0 on success or a negative ZO_E* code.  in_used/out_len optional. */
enum {
    ZO_OK = 0, ZO_E_TRUNC = -1, ZO_E_BTYPE = -2, ZO_E_NLEN = -3,
    ZO_E_CODELENS = -4, ZO_E_SYMBOL = -5, ZO_E_DIST = -6, ZO_E_OUTCAP = -7,
    ZO_E_HEADER = -8, ZO_E_CHECKSUM = -9, ZO_E_SIZE = -10
};
int zo_inflate_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                   size_t *out_len, size_t *in_used);
int zo_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                    size_t *out_len);
/* decodes all concatenated members */
int zo_gunzip(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
              size_t *out_len);
/* counts blocks by type while decoding a raw stream: counts[0..2] */
/* literal and match token counts of the last zo_inflate_raw* call (diagnostics) */
void zo_inflate_last_tokens(uint32_t out[2]);
/* dynamic blocks of the next zo_inflate_raw* calls are recorded into blocks[0..cap) (NULL stops it):
 * code lengths from the header, symbol counts from the decoded tokens */
typedef struct {
    uint8_t litlen_len[288], dist_len[32];
    uint32_t litlen_freq[288], dist_freq[32];
} zo_block_stats;
void zo_inflate_record_blocks(zo_block_stats *blocks, int cap);
int zo_inflate_recorded_blocks(void);
int zo_inflate_raw_stats(const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                         size_t *out_len, size_t *in_used, uint32_t counts[3]);

/* ---- test data: MT19937 as vendor/mt.js/mt.es.js:84-152 (init_genrand) -- */
typedef struct { uint32_t mt[624]; int idx; } zo_mt;
void zo_mt_seed(zo_mt *s, uint32_t seed);
uint32_t zo_mt_next(zo_mt *s);
/* test/browser.es/staff/util.js:87-134 generators (seed given as the JS
 * number's low 32 bits, mt.es.js:87) */
void zo_gen_random(uint8_t *dst, size_t n, uint32_t seed);
void zo_gen_sequential(uint8_t *dst, size_t n);
void zo_gen_random_sequential(uint8_t *dst, size_t n, uint32_t seed);
/* Zipf-like word text used by this repo's benches (integer-only LCG so the
 * JS generator in tests/golden/gen_golden.mjs emits identical bytes). */
void zo_gen_text(uint8_t *dst, size_t n, uint32_t seed);
/* twins of genRuns / 'small-alpha' in tests/golden/gen_golden.mjs */
void zo_gen_runs(uint8_t *dst, size_t n, uint32_t seed);
void zo_gen_small_alpha(uint8_t *dst, size_t n, uint32_t seed, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
