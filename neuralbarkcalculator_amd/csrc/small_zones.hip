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
//   border_merge   pixels whose W/NW/N/NE neighbour lies in another tile unite the two roots
//                  (atomicMin union-find on global memory with intermediate pointer jumping);
//   flatten_count  every mask pixel resolves its final root and adds one to that root's size;
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
    size[img + p] = 0;
  }
  const int t = threadIdx.x;
  s[t] = m ? t : -1;
  __syncthreads();
  if (m) {
    if (tx > 0 && s[t - 1] >= 0) unite_lds(s, t, t - 1);
    if (ty > 0) {
      if (s[t - TILE] >= 0) unite_lds(s, t, t - TILE);
      if (tx > 0 && s[t - TILE - 1] >= 0) unite_lds(s, t, t - TILE - 1);
      if (tx < TILE - 1 && s[t - TILE + 1] >= 0) unite_lds(s, t, t - TILE + 1);
    }
  }
  __syncthreads();
  if (inside) {
    int g = -1;
    if (m) {
      const int r = find_root_lds(s, t);
      g = (blockIdx.y * TILE + (r >> 5)) * W + blockIdx.x * TILE + (r & (TILE - 1));
    }
    parent[img + p] = g;                      // index inside the image, -1 outside the mask
  }
}

__global__ void border_merge_kernel(int* __restrict__ parent, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int* L = parent + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  if (L[p] < 0) return;
  const bool left_edge = (x & (TILE - 1)) == 0, top_edge = (y & (TILE - 1)) == 0, right_edge = (x & (TILE - 1)) == TILE - 1;
  if (left_edge && x > 0 && L[p - 1] >= 0) unite(L, p, p - 1);
  if (y > 0) {
    if (top_edge && L[p - W] >= 0) unite(L, p, p - W);
    if ((top_edge || left_edge) && x > 0 && L[p - W - 1] >= 0) unite(L, p, p - W - 1);
    if ((top_edge || right_edge) && x < W - 1 && L[p - W + 1] >= 0) unite(L, p, p - W + 1);
  }
}

__global__ void flatten_count_kernel(int* __restrict__ parent, int* __restrict__ size, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  int* L = parent + (size_t)blockIdx.z * H * W;
  int* S = size + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  int r = -1;
  if (x < W && L[p] >= 0) {
    r = find_root(L, p);                      // no links change any more: r is the component's root
    L[p] = r;                                 // shortcut for apply_kernel (which still walks: see there)
  }
  // One atomic per distinct root per wave instead of one per pixel: the background of a real label map
  // is a single component of ~10^6 pixels, and that many atomics on one address take milliseconds.
  unsigned long long todo = __ballot(r >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lr = __shfl(r, leader, 64);
    const unsigned long long same = __ballot(r == lr) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&S[lr], (int)__popcll(same));
    todo &= ~same;
  }
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
    // parent[q] may be any ancestor (pointer jumping of other threads can land after the pixel's own
    // final store in flatten_count): walk the few remaining hops, read-only
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
  if (phase == 1 && counts) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      c1 += __shfl_xor(c1, off, 64);
      c2 += __shfl_xor(c2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      if (c1) atomicAdd(&counts[(size_t)blockIdx.z * 3 + 1], (unsigned long long)c1);
      if (c2) atomicAdd(&counts[(size_t)blockIdx.z * 3 + 2], (unsigned long long)c2);
    }
  }
}

__global__ void finish_counts_kernel(unsigned long long* counts, int N, unsigned long long pixels) {
  const int i = threadIdx.x;
  if (i < N) counts[i * 3] = pixels - counts[i * 3 + 1] - counts[i * 3 + 2];
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
    hipLaunchKernelGGL(border_merge_kernel, rows, dim3(256), 0, s, parent, H, W);
    hipLaunchKernelGGL(flatten_count_kernel, rows, dim3(256), 0, s, parent, size, H, W);
    hipLaunchKernelGGL(apply_kernel<LabelT>, rows, dim3(256), 0, s, labels, bg, parent, size, H, W, min_pixels, phase,
                       exclude_nodes, counts);
  }
  if (counts) hipLaunchKernelGGL(finish_counts_kernel, dim3(1), dim3(256), 0, s, counts, N, (unsigned long long)H * W);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_remove_small_zones(void* labels, int labels_i64, int N, int H, int W, int min_pixels, int exclude_nodes,
                                     unsigned char* bg, int* parent, int* size, unsigned long long* counts, hipStream_t s) {
  if (N < 1 || N > 85 || H < 1 || W < 1 || H > 65535 || (long long)H * W > 0x7fffffffLL) return hipErrorInvalidValue;
  if (labels_i64) return run(static_cast<long long*>(labels), N, H, W, min_pixels, exclude_nodes, bg, parent, size, counts, s);
  return run(static_cast<unsigned char*>(labels), N, H, W, min_pixels, exclude_nodes, bg, parent, size, counts, s);
}

}  // namespace nbc
