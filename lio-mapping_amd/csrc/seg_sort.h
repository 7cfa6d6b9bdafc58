// seg_sort.h — stable LSD radix sort of (32-bit key, 32-bit value) pairs inside SEGMENTS of one array: every window of a batch sorts
// its own range (a window's points are contiguous: no window bits in the key), all windows in the same launches.
//
// Used by the batched BuildLocalMap (Estimator.cc:1518-1519: pcl::VoxelGrid orders its points by voxel index — the sort's key — and sums
// a voxel's points in their original order, which a STABLE sort keeps) and by the K-NN grid (Estimator.cc:1544-1545: points ordered by
// cell).  It replaces rocprim::radix_sort_pairs over 64-bit (window | key) keys: 9-bit digits over the key bits a window really uses
// (three passes for 27 bits instead of five 8-bit passes over 40), 8 B per element and direction instead of 12.
//
// One pass = three launches over (tiles of the largest segment, segments):
//   k_ss_hist     a tile's digit histogram (LDS atomics)                                    reads 4 B per element
//   k_ss_scan     one block per segment: exclusive scan of its (digit, tile) counts — digit-major, so a digit's tiles are consecutive
//   k_ss_scatter  the tile again: a wave owns consecutive rounds of 64 elements, ranks them among equal digits with ballots, keeps a
//                 running per-wave count in LDS; the waves' counts are prefixed per digit behind one barrier; scatter.  reads 8, writes 8
// Stable by construction: tiles, waves, rounds and lanes are all walked in index order.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lio {

#define SS_ITEMS 16        // elements per thread
#define SS_MAX_BITS 9      // digit width of a pass: up to 512 bins

struct SegDesc {
  int off;        // first element of the segment in the arrays
  int n;          // elements (the sorted output occupies [off, off + n))
  int hist_off;   // first entry of its (digit, tile) table in the histogram array: (tiles of all segments before it) << bits
};

// how a pass sees a key.  mode 0: as stored.  mode 1 (first pass of the voxel sort): the stored key is PCL's voxel index in absolute
// cells, z (9 bits, + 256) | y (11, + 1024) | x (11, + 1024), all ones = no point; the sort runs on the key relative to the window's
// own bounds, z' | y' | x' packed to the bits the window uses (KeyLayout), and writes THAT key out.
struct KeyLayout { int mx, my, mz; int bx, by; int bits; };   // mins in the stored key's offset space; shifts; total bits (> 27: not sortable in three 9-bit passes)

struct SegSortPlan {
  int threads;          // 256, 512 (or 1024) per block: tile = threads * SS_ITEMS
  int max_tiles;        // tiles of the largest segment
  size_t hist_entries;  // total (digit, tile) entries of one pass: sum over segments of tiles << bits
};
// tile size by the size of the launch; fills desc[k].hist_off.  sizes[k] = elements of segment k.
SegSortPlan seg_sort_plan(SegDesc *desc, int nseg, int bits_per_pass);

// One pass on digit [shift, shift + bits) of the keys.  layout != nullptr: mode 1 (per-segment layouts, device array); vals_in == nullptr:
// the values are the elements' own positions in the array (first pass).  hist: plan.hist_entries uint32 of scratch.
void seg_sort_pass(const SegDesc *d_desc, int nseg, const SegSortPlan &plan, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                   uint32_t *hist, int shift, int bits, const KeyLayout *layout, hipStream_t s);

// test hook (lio_seg_sort_pairs): sorts host arrays through `passes` passes of `bits` bits; returns false on bad arguments
bool seg_sort_host_test(const uint32_t *keys, const uint32_t *vals_or_null, size_t n_total, const int *seg_off, const int *seg_n, int nseg, int bits, int passes,
                        uint32_t *keys_out, uint32_t *vals_out);

}  // namespace lio
