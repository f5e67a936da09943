/* GPU round-trip verifier for raw DEFLATE streams -- a test accelerator, not
 * part of the compression product and not loaded by it.
 *
 * Decoding stays with the reference's JavaScript inflaters for users
 * (src/rawinflate.js:135-416); this library only lets the tests and the
 * benchmark check EVERY unit of a full-size batch on the device instead of a
 * sample on the host.  It decodes each unit's stream strictly (RFC 1951:
 * block types, LEN/NLEN, HLIT/HDIST ranges, over-subscribed or incomplete
 * code sets, symbols 286/287 and distance codes 30/31, distances beyond the
 * bytes produced) and compares the result with the original bytes on the fly:
 * a literal must equal the original byte, a match (length, distance) must
 * satisfy orig[p+i] == orig[p+i-distance] -- by induction the decoded output
 * equals the original without being stored.
 */
#ifndef ZGPU_VERIFY_H
#define ZGPU_VERIFY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* per-unit status written by zgpu_verify_streams */
enum {
    ZGPU_V_OK = 0,
    ZGPU_V_TRUNCATED = 1,   /* ran out of compressed bytes */
    ZGPU_V_BLOCK_TYPE = 2,  /* BTYPE 3 */
    ZGPU_V_STORED = 3,      /* LEN != ~NLEN */
    ZGPU_V_HEADER = 4,      /* HLIT > 286, HDIST > 30, bad repeat, no end-of-block code */
    ZGPU_V_CODESET = 5,     /* over-subscribed or incomplete code */
    ZGPU_V_SYMBOL = 6,      /* undecodable code, literal/length 286-287, distance 30-31 */
    ZGPU_V_DISTANCE = 7,    /* distance beyond the bytes produced so far */
    ZGPU_V_MISMATCH = 8,    /* decoded byte differs from the original */
    ZGPU_V_LENGTH = 9,      /* decoded size differs from the original size */
    ZGPU_V_TRAILING = 10,   /* compressed bytes left over / stream not ended as flagged */
};

/* flags: bit 0 = every unit must end with BFINAL (independent streams);
 * otherwise a unit must end with an empty stored block (chunk of one stream)
 * except the last, which must end with BFINAL.  bit 1 = the units are
 * consecutive pieces of one buffer and a unit's matches may reach back into
 * the pieces before it (window priming).
 * All pointers are device pointers; comp_off / orig_off have n_units + 1
 * entries; status has n_units.  stream: hipStream_t (NULL = default).
 * Returns 0, or the negated hipError_t of a failed launch. */
int zgpu_verify_streams(const void *d_comp, const uint64_t *d_comp_off, const void *d_orig,
                        const uint64_t *d_orig_off, uint32_t n_units, uint32_t flags, uint32_t *d_status,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif
