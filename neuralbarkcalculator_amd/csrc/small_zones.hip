// remove_small_zones on the GPU: /root/reference/src/bark_calculator/utils.py:135-148, called at
// models.py:271 between the argmax and the statistics.  Byte/index work, bound by HBM/L2 traffic.
//
// With m = (labels == 0) (the "Nothing" mask), skimage semantics (connectivity=2 = 8-neighbourhood):
//   1. remove_small_holes(m, 150):   8-connected components of ~m smaller than 150 px join m;
//   2. remove_small_objects(m, 150): 8-connected components of the filled m smaller than 150 px leave m;
//   3. pixels that left m and were class 0 become class 1; pixels that joined m become class 0.
// Both steps are "label the 8-connected components of a binary mask, drop the small ones":
//   tile_label     one 32x32 tile per block: union-find in LDS over the tile's mask pixels (each pixel
//                  with its W, NW, N, NE neighbours), every pixel then points at the tile-local root
//                  (a global pixel index), so chains in global memory only ever link tile roots;
//   border_rows/cols  pixels on a tile border unite their root with the touching runs across it
//                  (atomicMin union-find on global memory with intermediate pointer jumping);
//   sum_sizes      every tile-local root adds its tile's pixel count to its final root;
//   apply          components below the threshold flip in the mask.
// The result (which pixels flip) depends only on component sizes, which are exact integers, so the
// output equals the CPU restatement (scipy.ndimage.label) bit for bit.
#include "nbc_kernels.hpp"

namespace nbc {
namespace {

constexpr int TILE = 32;

// Root of x with intermediate pointer jumping: every node passed is re-pointed at its grandparent.
// Parents only ever decrease and always stay inside the node's (eventual) component, so the plain
// stores race benignly with the atomicMin links of unite() (ECL-CC's "representative").
__device__ __forceinline__ int find_root(int* L, int x) {
  int curr = L[x];
  if (curr != x) {
    int prev = x, next;
    while (curr > (next = L[curr])) {
      L[prev] = next;
      prev = curr;
      curr = next;
    }
  }
  return curr;
}
__device__ __forceinline__ int find_root_lds(const volatile int* L, int x) {
  int p = L[x];
  while (p != x) { x = p; p = L[x]; }
  return x;
}
// link the larger root under the smaller one; retried when another thread re-rooted it meanwhile
__device__ __forceinline__ void unite(int* L, int a, int b) {
  while (true) {
    a = find_root(L, a);
    b = find_root(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;
  }
}
__device__ __forceinline__ void unite_lds(int* L, int a, int b) {
  while (true) {
    a = find_root_lds(L, a);
    b = find_root_lds(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;
  }
}

// phase 0: mask = (label != 0) is labelled ("holes" of the Nothing mask); phase 1: mask = bg.
// bg[] holds the Nothing mask (1 = background); written here in phase 0.
template <typename LabelT>
__global__ __launch_bounds__(TILE* TILE) void tile_label_kernel(const LabelT* __restrict__ labels, unsigned char* __restrict__ bg,
                                                                 int* __restrict__ parent, int* __restrict__ size,
                                                                 int H, int W, int phase) {
  __shared__ int s[TILE * TILE];
  __shared__ int cnt[TILE * TILE];
  const int tx = threadIdx.x & (TILE - 1), ty = threadIdx.x >> 5;
  const int x = blockIdx.x * TILE + tx, y = blockIdx.y * TILE + ty;
  const size_t img = (size_t)blockIdx.z * H * W;
  const bool inside = x < W && y < H;
  const int p = y * W + x;
  bool m = false;
  if (inside) {
    if (phase == 0) {
      const bool is_bg = labels[img + p] == 0;
      bg[img + p] = is_bg ? 1 : 0;
      m = !is_bg;
    } else {
      m = bg[img + p] != 0;
    }
  }
  const int t = threadIdx.x;
  // Horizontal runs first, without atomics: a wave holds two rows of the tile; a mask pixel points at
  // the first pixel of its run (the position after the nearest unset bit to its left).
  const unsigned row_bits = (unsigned)(__ballot(m) >> (32 * ((t >> 5) & 1)));
  const unsigned zeros_left = ~row_bits & ((1u << tx) - 1u);
  const int run_start = zeros_left ? 32 - __clz((int)zeros_left) : 0;
  s[t] = m ? (t - tx + run_start) : -1;
  cnt[t] = 0;
  __syncthreads();
  // Then one union per pair of touching runs of consecutive rows: the leftmost pixel of the overlap
  // links upwards (N, or NW / NE when only a diagonal touches); the rest of the run is already joined.
  if (m && ty > 0) {
    const bool w = tx > 0 && ((row_bits >> (tx - 1)) & 1u), e = tx < TILE - 1 && ((row_bits >> (tx + 1)) & 1u);
    const bool n = s[t - TILE] >= 0;
    const bool nw = tx > 0 && s[t - TILE - 1] >= 0, ne = tx < TILE - 1 && s[t - TILE + 1] >= 0;
    if (n) {
      if (!(w && nw)) unite_lds(s, t, t - TILE);
    } else {
      if (nw && !w) unite_lds(s, t, t - TILE - 1);
      if (ne && !e) unite_lds(s, t, t - TILE + 1);
    }
  }
  __syncthreads();
  const int r = m ? find_root_lds(s, t) : -1;
  // pixels per tile-local root, one LDS atomic per distinct root per wave
  unsigned long long todo = __ballot(r >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lr = __shfl(r, leader, 64);
    const unsigned long long same = __ballot(r == lr) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&cnt[lr], (int)__popcll(same));
    todo &= ~same;
  }
  __syncthreads();
  if (inside) {
    parent[img + p] = m ? (blockIdx.y * TILE + (r >> 5)) * W + blockIdx.x * TILE + (r & (TILE - 1)) : -1;   // index inside the image
    size[img + p] = (m && r == t) ? cnt[t] : 0;   // non-zero exactly at the tile-local roots
  }
}

// Unions across tile borders.  Only border pixels have anything to do, and the run rule of the tile
// kernel applies across borders too, so the launch covers just them:
//   rows  y = 32k (k >= 1): a pixel links to the row above through N unless its W neighbour already does
//         (W and NW both set), or through NW / NE when only a diagonal touches;
//   columns x = 32k (k >= 1): the pixel right of the border links to W; the diagonals across a vertical
//         border are needed only when neither the straight neighbour of the pixel's own tile (N) nor
//         the pixel across the border in its own row (W resp. E) is set: otherwise that neighbour's own
//         W-link already joins the two.
__global__ void border_rows_kernel(int* __restrict__ parent, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = (blockIdx.y + 1) * TILE;
  if (x >= W || y >= H) return;
  int* L = parent + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  if (L[p] < 0) return;
  const bool n = L[p - W] >= 0;
  const bool w = x > 0 && L[p - 1] >= 0, e = x < W - 1 && L[p + 1] >= 0;
  const bool nw = x > 0 && L[p - W - 1] >= 0, ne = x < W - 1 && L[p - W + 1] >= 0;
  if (n) {
    if (!(w && nw)) unite(L, p, p - W);
  } else {
    if (nw && !w) unite(L, p, p - W - 1);
    if (ne && !e) unite(L, p, p - W + 1);
  }
}
__global__ void border_cols_kernel(int* __restrict__ parent, int H, int W) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x, x = (blockIdx.y + 1) * TILE;      // x: first column right of a border
  if (y >= H || x >= W) return;
  int* L = parent + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;                    // right of the border; p - 1 is left of it
  const bool top = (y & (TILE - 1)) == 0;     // rows on a horizontal border are handled by border_rows_kernel
  const bool r_set = L[p] >= 0, l_set = L[p - 1] >= 0;
  if (r_set && l_set) unite(L, p, p - 1);
  if (y == 0 || top) return;
  if (r_set && !l_set && L[p - W] < 0 && L[p - W - 1] >= 0) unite(L, p, p - W - 1);          // NW across the border
  if (l_set && !r_set && L[p - 1 - W] < 0 && L[p - W] >= 0) unite(L, p - 1, p - W);          // NE across the border
}

// Component sizes: tile_label_kernel left the pixel count of every tile-local root in size[]; a local
// root that was linked under another root adds its count to its final root.  Only the (few thousand)
// local roots do anything here; their atomics spread over as many final roots as there are components.
__global__ void sum_sizes_kernel(int* __restrict__ parent, int* __restrict__ size, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int* L = parent + (size_t)blockIdx.z * H * W;
  int* S = size + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  const int mine = S[p];
  if (mine == 0) return;                      // not a tile-local root
  const int r = find_root(L, p);              // no links change any more: r is the component's root
  if (r != p) atomicAdd(&S[r], mine);
}

// phase 0: small components of ~bg join bg.  phase 1: small components of bg leave it, then the labels
// are rewritten (utils.py:145-146), optionally remapped 2 -> 1 (models.py:273-276) and counted.
template <typename LabelT>
__global__ void apply_kernel(LabelT* __restrict__ labels, unsigned char* __restrict__ bg, const int* __restrict__ parent,
                             const int* __restrict__ size, int H, int W, int min_pixels, int phase, int exclude_nodes,
                             unsigned long long* __restrict__ counts) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  const size_t img = (size_t)blockIdx.z * H * W;
  unsigned c1 = 0, c2 = 0;
  if (x < W) {
    const size_t q = img + (size_t)y * W + x;
    // parent[q] is the pixel's tile-local root (or, after pointer jumping, an ancestor of it): walk the
    // few remaining hops to the component's root, read-only
    int r = parent[q];
    if (r >= 0) {
      const int* L = parent + img;
      int up = L[r];
      while (up != r) { r = up; up = L[r]; }
    }
    const bool small = r >= 0 && size[img + r] < min_pixels;
    if (phase == 0) {
      if (small) bg[q] = 1;
    } else {
      const bool kept = bg[q] != 0 && !small;
      int v = (int)labels[q];
      if (!kept && v == 0) v = 1;
      if (kept && v != 0) v = 0;
      if (exclude_nodes && v == 2) v = 1;
      labels[q] = (LabelT)v;
      c1 = v == 1; c2 = v == 2;
    }
  }
  if (phase == 1 && counts) {              // one global atomic per class and block
    __shared__ unsigned blk[2];
    if (threadIdx.x < 2) blk[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      c1 += __shfl_xor(c1, off, 64);
      c2 += __shfl_xor(c2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      if (c1) atomicAdd(&blk[0], c1);
      if (c2) atomicAdd(&blk[1], c2);
    }
    __syncthreads();
    if (threadIdx.x < 2 && blk[threadIdx.x])
      atomicAdd(&counts[(size_t)blockIdx.z * 3 + 1 + threadIdx.x], (unsigned long long)blk[threadIdx.x]);
  }
}

__global__ void finish_counts_kernel(unsigned long long* counts, int N, unsigned long long pixels) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
    counts[(size_t)i * 3] = pixels - counts[(size_t)i * 3 + 1] - counts[(size_t)i * 3 + 2];
}

template <typename LabelT>
hipError_t run(LabelT* labels, int N, int H, int W, int min_pixels, int exclude_nodes, unsigned char* bg, int* parent,
               int* size, unsigned long long* counts, hipStream_t s) {
  const dim3 tiles((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, N);
  const dim3 rows((W + 255) / 256, H, N);
  if (counts) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(unsigned long long) * 3 * N, s);
    if (e != hipSuccess) return e;
  }
  for (int phase = 0; phase < 2; ++phase) {
    hipLaunchKernelGGL(tile_label_kernel<LabelT>, tiles, dim3(TILE * TILE), 0, s, labels, bg, parent, size, H, W, phase);
    if (H > TILE) hipLaunchKernelGGL(border_rows_kernel, dim3((W + 255) / 256, (H - 1) / TILE, N), dim3(256), 0, s, parent, H, W);
    if (W > TILE) hipLaunchKernelGGL(border_cols_kernel, dim3((H + 255) / 256, (W - 1) / TILE, N), dim3(256), 0, s, parent, H, W);
    hipLaunchKernelGGL(sum_sizes_kernel, rows, dim3(256), 0, s, parent, size, H, W);
    hipLaunchKernelGGL(apply_kernel<LabelT>, rows, dim3(256), 0, s, labels, bg, parent, size, H, W, min_pixels, phase,
                       exclude_nodes, counts);
  }
  if (counts) hipLaunchKernelGGL(finish_counts_kernel, dim3((N + 255) / 256), dim3(256), 0, s, counts, N, (unsigned long long)H * W);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_remove_small_zones(void* labels, int labels_i64, int N, int H, int W, int min_pixels, int exclude_nodes,
                                     unsigned char* bg, int* parent, int* size, unsigned long long* counts, hipStream_t s) {
  if (N < 1 || N > 65535 || H < 1 || W < 1 || H > 65535 || (long long)H * W > 0x7fffffffLL) return hipErrorInvalidValue;
  if (labels_i64) return run(static_cast<long long*>(labels), N, H, W, min_pixels, exclude_nodes, bg, parent, size, counts, s);
  return run(static_cast<unsigned char*>(labels), N, H, W, min_pixels, exclude_nodes, bg, parent, size, counts, s);
}

}  // namespace nbc
