// Shared between the device kernels and the host side of libzgpu.
#pragma once
#include <stdint.h>

// tokens one workgroup can hold for one DEFLATE block (64 KiB of input can
// yield at most 65536 tokens, plus end-of-block and the final literal)
#define ZGPU_TOK_CAP (65536u + 64u)
// workgroups of the DEFLATE kernel that fit one CU (LDS-bound)
#define ZGPU_DEFLATE_WG_PER_CU 1

// checksum kernel geometry: every workgroup folds ZGPU_CK_SEG bytes
#define ZGPU_CK_THREADS 256
#define ZGPU_CK_SEG (1u << 20)
