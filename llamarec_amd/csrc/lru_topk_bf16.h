// lru_topk_bf16.h -- the two APPROXIMATE (bf16 MFMA) passes of the top-K's bound -> candidates -> exact rescoring path
// (lru_topk.hip explains the proof; lru_topk_bf16.hip holds the kernels, compiled with -fno-honor-nans: see there).
#pragma once
#include "lr_common.h"

#define TK_CAND_CAP 1024  // candidate slots per user (1 M items: ~270 above the bound + ~40 % in the 2 delta band)
#define TK_CAND_LIST_SLOTS 20480   // 16-bit candidate slots in a workgroup's LDS lists (40 KiB; with the stage ring and the
                                   // counters <= 77 KiB: two workgroups per CU): 40 per user at 512 users per workgroup, 20
                                   // at 1 024. An entry is an offset from the chunk's first item, so a chunk is at most
                                   // lr_bf16_max_chunk_tiles() tiles: ~1 000 candidates per user at 1 M items are then ~16
                                   // (8) per chunk at worst and ~5 (3) typically; entries past a full list go to the global list one atomic each

// users per workgroup of the two bf16 passes (8 waves x 2 MFMA column tiles of 32 users; 4 were measured and dropped) and the longest chunk of
// tiles the candidate pass's per-chunk lists are sized for, both functions of the batch size only
int lr_bf16_users_per_wg(int B);
static inline int lr_bf16_max_chunk_tiles(int B) { return lr_bf16_users_per_wg(B) >= 1024 ? 256 : 512; }   // x 32 items < 65 536

struct BoundParams {
  const unsigned short* emb16;  // bf16 table in A-fragment order [tile][step][lane][8] (lr_lru_pack)
  const float* bias;            // [rows_padded]
  const float* bias_tail;       // [32] the last tile's biases, -inf on padding rows (LrLruLayout::item_stats + 32)
  int n_rows, n_tiles;
  const float* q;               // [B][64]
  int B;
  float* tmax;                  // [B][ld]: maxima of GROUPS of 2^gshift consecutive tiles
  int ld;                       // number of groups rounded up to 4
  int tiles_per_chunk;          // multiple of max(4, 2^gshift)
  int gshift;                   // 0: one maximum per tile (catalogs up to 65 536 items); >= 2: per 4, 8, 16 .. tiles
};

struct CandParams {
  const unsigned short* emb16;
  const float* bias;
  const float* bias_tail;
  int n_tiles;
  const float* q;
  int B;
  const float* cand_thresh;  // [B]
  int* cand_count;           // [B]
  int32_t* cand;             // [B][TK_CAND_CAP] item ids
  int tiles_per_chunk;       // <= lr_bf16_max_chunk_tiles(B)
};

// grid = (chunks, user tiles of lr_bf16_users_per_wg(B)); both passes must be launched with the same q, table and B (same
// operands -> the same approximate scores)
int lr_launch_item_bound(const BoundParams& p, int chunks, hipStream_t st);
int lr_launch_item_cand(const CandParams& p, int chunks, hipStream_t st);
