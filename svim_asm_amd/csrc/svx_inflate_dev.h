// svx_inflate_dev.h — what svx_bam.cpp (the device leg of the sequence slices) and svx_inflate.hip (the kernels) agree on.
#pragma once
#include <stdint.h>

// slots (8 bytes: destination | length << 16, distance) of a member's token list in the two-pass forms: a BGZF member is
// 65 536 bytes at most and a match yields 3 at least
#define SVX_INFLATE_TOK_STRIDE (65536u / 3u + 2u)
// members whose token lists the arena holds at once (175 KB each: 3.6 GB at most); a call's members go out that many at a
// time, one slice of launches behind the other.  A slice costs at least one member's latency (3.6 ms), so the arena is as
// large as a full-size sample's call (11 k members a reader under its sequence slices + 6.5 k the record walks touched):
// 7 261 members 8.1 ms in one slice, 11.2 in slices of 6 144; 28 000 members 27 ms in slices of 16 384, 34.5 in slices of
// 6 144 (profiles/README.md, round 6)
#ifndef SVX_INFLATE_ARENA_MEMBERS
#define SVX_INFLATE_ARENA_MEMBERS 20480u
#endif

// svx_bgzf_inflate_on_stream / svx_gather_ranges_on_stream (svx_inflate.hip), reached through pointers: svx_bam.cpp also
// builds alone, without the kernels, for the CPU sanitizer tests.  hipError_t as int.
typedef int (*svx_inflate_launch_fn)(void* stream, const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                     const uint32_t* d_isize, const uint32_t* d_crc, uint32_t n_members, uint8_t* d_out,
                                     const uint64_t* d_out_off, uint32_t* d_status, uint32_t* d_n_tok, void* d_tok, uint32_t tok_members);
typedef int (*svx_gather_launch_fn)(void* stream, const uint8_t* d_src, const uint64_t* d_src_off, const uint32_t* d_len,
                                    const uint64_t* d_dst_off, uint32_t n, uint8_t* d_dst);
