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
  int n_chunks, n_user_groups;  // the launch's geometry (set by lr_launch_item_bound): workgroup -> (chunk, user group), below
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
  int n_chunks, n_user_groups;   // set by lr_launch_item_cand
};

// Workgroup -> (chunk of tiles, group of lr_bf16_users_per_wg(B) users). Every user group streams the WHOLE packed table, so the
// groups that read one chunk must read it together and through ONE L2: with a (chunks, groups) grid the same chunk came back
// `chunks` workgroups later and on any XCD -- 8 groups x 128 MB crossed the fabric per pass at 4 096 users (L2 hit rate 0.12 /
// 0.21, measured in round 4). Now a 1-D grid: workgroup id -> XCD id & 7 (round-robin dispatch), slot id >> 3; the slots of an
// XCD walk chunk-major -- chunk = xcd + 8 (slot / G), group = slot % G -- so the G workgroups of a chunk are dispatched back
// to back on the same XCD and the chunk crosses the fabric once.
struct TkWho {
  int chunk, group;
};
__device__ __forceinline__ TkWho tk_who(int n_chunks, int n_user_groups) {
  const int id = blockIdx.x, slot = id >> 3;
  TkWho w;
  w.chunk = (id & 7) + 8 * (slot / n_user_groups);
  w.group = slot % n_user_groups;
  if (w.chunk >= n_chunks) w.chunk = -1;
  return w;
}
static inline unsigned tk_grid(int n_chunks, int n_user_groups) { return 8u * (unsigned)((n_chunks + 7) / 8) * (unsigned)n_user_groups; }
// both passes must be launched with the same q, table and B (same operands -> the same approximate scores)
int lr_launch_item_bound(const BoundParams& p, int chunks, hipStream_t st);
int lr_launch_item_cand(const CandParams& p, int chunks, hipStream_t st);
