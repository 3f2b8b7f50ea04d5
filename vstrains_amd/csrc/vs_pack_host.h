// Host-side 2-bit packing of a sequence line (see vs_pack_host.cpp); flags as in vs_internal.h
// (VS_FLAG_N / VS_FLAG_INVALID) plus 0x80 for a byte >= 0x80.
#pragma once
#include <stdint.h>
#define VS_PACK_FLAG_N 1u
#define VS_PACK_FLAG_INVALID 2u
#define VS_PACK_FLAG_NON_ASCII 0x80u
uint32_t vs_pack_sequence_host(const uint8_t *q, uint32_t len, uint32_t *out);        // vector body picked at load time
uint32_t vs_pack_sequence_host_plain(const uint8_t *q, uint32_t len, uint32_t *out);  // byte by byte (the check for the other)
