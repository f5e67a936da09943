// Shared between the device kernels and the host side of libzgpu.
#pragma once
#include <stdint.h>

// One DEFLATE block per this many input bytes.  The token stream between the two
// kernels is made of 16-bit slots: a literal or end-of-block (256) takes one, a match
// two (0x8000 | length - 3, then distance - 1), so a block yields at most one slot per
// input byte plus end-of-block; block b of a unit starts ZGPU_BLOCK_TOK * b slots into
// the unit's region (a multiple of 4, so four slots load as one 8 bytes)
#define ZGPU_BLOCK_BYTES 65536u
#define ZGPU_BLOCK_TOK (ZGPU_BLOCK_BYTES + 4u)
// workgroups of the DEFLATE kernel that fit one CU (LDS-bound)
#define ZGPU_DEFLATE_WG_PER_CU 1
// workgroups of the emit kernel queued per CU (small LDS footprint, latency bound)
#define ZGPU_EMIT_WG_PER_CU 8

// checksum kernel geometry: every workgroup folds ZGPU_CK_SEG bytes
#define ZGPU_CK_THREADS 256
#define ZGPU_CK_SEG (1u << 20)
// wide checksum kernel (large buffers): one 1024-thread workgroup per CU, 4 MiB segments
#define ZGPU_CKW_THREADS 1024
#define ZGPU_CKW_SEG (4u << 20)
#define ZGPU_CKW_MIN_BYTES (64ull << 20)
// consecutive 16-byte pieces a lane folds per round (1, 2 or 4), and the entries of the x^(128 * 2^lv) table
#ifndef ZGPU_CKW_PIECES
#define ZGPU_CKW_PIECES 2
#endif
#define ZGPU_CKW_XPOW 12
