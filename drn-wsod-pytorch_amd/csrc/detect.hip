// Inference tail for gfx950: score threshold + compaction in row-major (nonzero) order, stable
// descending sort, class-offset greedy NMS with early exit at top-k.  Indices must be bit-exact
// vs the CPU path, so every float op follows the oracle's order (built with -ffp-contract=off).
//
// Replaces fast_rcnn_inference_single_image (projects/WSL/wsl/modeling/roi_heads/fast_rcnn.py:88-141)
// and batched_nms (detectron2/layers/nms.py:10-29 over torchvision nms / batched_nms).
//
// Design: the reference sorts and suppresses ALL candidates (up to R*K = 40k-320k) and then keeps
// keep[:100].  Greedy NMS is order-causal: whether candidate i survives depends only on survivors
// before it, so we walk the sorted list 64 candidates (one wave) at a time against the <= topk
// survivors held in LDS and stop as soon as topk are kept - identical output, O(n*topk) work.
#include "drn_common.h"

namespace {

struct CandParams {
  const float* boxes;   // [R][4*nreg]
  const float* scores;  // [R][K+1]
  int R, K, nreg;
  float img_h, img_w, thresh;
  float* c_box;   // [cap][4] clipped boxes
  float* c_score; // [cap]
  int* c_row; int* c_cls;  // [cap]
  int* count;     // [1]
  float* maxcoord;  // [1] max over candidate box coordinates
  int cap;
  int* rowmap;    // [R] scratch: index of row r among the finite rows, or -1
};

// finite-row filter (fast_rcnn.py:108-111), one wave per row over many blocks: rowmap[r] = 0 (all K+1 scores and 4*nreg box
// coordinates finite) or -1.  Inside the single-block kernel below this was 101 loads per thread, each waiting for the previous one's
// verdict (`&& finite`), from ONE CU: ~120 of that kernel's 150 us (profiles/r5_63_infer480_kernel_stats.txt).
__global__ __launch_bounds__(256) void rows_finite_kernel(CandParams p) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.R) return;
  const float* srow = p.scores + (long)r * (p.K + 1);
  const float* brow = p.boxes + (long)r * 4 * p.nreg;
  bool finite = true;
  for (int k = lane; k <= p.K; k += 64) finite = finite && isfinite(srow[k]);
  for (int k = lane; k < 4 * p.nreg; k += 64) finite = finite && isfinite(brow[k]);
  const bool all = __ballot(!finite) == 0;
  if (lane == 0) p.rowmap[r] = all ? 0 : -1;
}

// single block: the rows that passed rows_finite_kernel, numbered in order - rows with any non-finite box or score are dropped BEFORE
// indexing (fast_rcnn.py:108-111), so every later row index refers to the compacted arrays
__global__ __launch_bounds__(1024) void rows_number_kernel(CandParams p) {
  __shared__ int wcnt[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    base = 0;
    p.maxcoord[0] = -INFINITY;
  }
  __syncthreads();
  for (int start = 0; start < p.R; start += 1024) {
    const int r = start + threadIdx.x;
    const bool finite = r < p.R && p.rowmap[r] >= 0;
    const unsigned long long bal = __ballot(finite);
    const int wprefix = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) wcnt[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int q = 0; q < 16; ++q) { if (q < w) woff += wcnt[q]; tot += wcnt[q]; }
    if (r < p.R) p.rowmap[r] = finite ? base + woff + wprefix : -1;
    __syncthreads();
    if (threadIdx.x == 0) base += tot;
    __syncthreads();
  }
}

// threshold + ordered compaction of the R x K scores (row-major = nonzero order) over many blocks: a tile of CAND_TILE consecutive
// entries per block, a thread takes CAND_EPT CONSECUTIVE entries (entry order = thread order = output order).  Pass 0 leaves
// every tile's count, pass 1 places a tile behind the sum of the counts in front of it - the output is a function of the input
// alone, as in the single-block form this replaces (150 us per image from one CU: 40 000 entries, 280 000 scattered 4-byte stores;
// profiles/r5_63_infer480_kernel_stats.txt).  The largest candidate coordinate is an atomic max on the bits of non-negative
// floats (the boxes are clipped to the image first): exact whatever the order.
constexpr int CAND_THREADS = 256, CAND_EPT = 8, CAND_TILE = CAND_THREADS * CAND_EPT;
template <int PASS>
__global__ __launch_bounds__(CAND_THREADS) void candidates_kernel(CandParams p, int* tile_cnt, int ntiles) {
  __shared__ int wcnt[CAND_THREADS / 64];
  __shared__ int sbase;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long total = (long)p.R * p.K;
  const long i0 = (long)blockIdx.x * CAND_TILE + (long)threadIdx.x * CAND_EPT;
  unsigned flags = 0;
  float sv[CAND_EPT];
  int n_mine = 0;
#pragma unroll
  for (int e = 0; e < CAND_EPT; ++e) {
    const long i = i0 + e;
    sv[e] = 0.f;
    if (i < total) {
      const int r = (int)(i / p.K), c = (int)(i - (long)r * p.K);
      sv[e] = p.scores[(long)r * (p.K + 1) + c];
      if (p.rowmap[r] >= 0 && sv[e] > p.thresh) flags |= 1u << e, ++n_mine;
    }
  }
  // inclusive prefix of n_mine over the wave's lanes, then over the block's waves
  int incl = n_mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) wcnt[w] = incl;
  if (PASS == 1 && threadIdx.x == 0) {
    int b = 0;
    for (int t = 0; t < (int)blockIdx.x; ++t) b += tile_cnt[t];
    sbase = b;
  }
  __syncthreads();
  int woff = 0, tot = 0;
  for (int q = 0; q < CAND_THREADS / 64; ++q) { if (q < w) woff += wcnt[q]; tot += wcnt[q]; }
  if (PASS == 0) {
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
    return;
  }
  int pos = sbase + woff + incl - n_mine;
  float mymax = -INFINITY;
#pragma unroll
  for (int e = 0; e < CAND_EPT; ++e)
    if (flags >> e & 1) {
      if (pos < p.cap) {
        const long i = i0 + e;
        const int r = (int)(i / p.K), c = (int)(i - (long)r * p.K);
        const float* b = p.boxes + (long)r * 4 * p.nreg + (p.nreg == 1 ? 0 : 4 * c);
        const float x1 = fminf(fmaxf(b[0], 0.f), p.img_w), y1 = fminf(fmaxf(b[1], 0.f), p.img_h);
        const float x2 = fminf(fmaxf(b[2], 0.f), p.img_w), y2 = fminf(fmaxf(b[3], 0.f), p.img_h);
        p.c_box[4 * (long)pos] = x1; p.c_box[4 * (long)pos + 1] = y1; p.c_box[4 * (long)pos + 2] = x2; p.c_box[4 * (long)pos + 3] = y2;
        p.c_score[pos] = sv[e]; p.c_row[pos] = p.rowmap[r]; p.c_cls[pos] = c;
        mymax = fmaxf(mymax, fmaxf(fmaxf(x1, y1), fmaxf(x2, y2)));
      }
      ++pos;
    }
  mymax = wave_max(mymax);
  // (clipped coordinates are >= 0, or NaN-free by the finite filter: their bit patterns order like the values; -inf stays for "none")
  if (lane == 0 && mymax >= 0.f) atomicMax((int*)p.maxcoord, __builtin_bit_cast(int, mymax));
  if ((int)blockIdx.x == ntiles - 1 && threadIdx.x == 0) {
    const int n = sbase + tot;
    p.count[0] = n < p.cap ? n : p.cap;
  }
}

// ---- stable descending sort of the candidates by score (round 3: own kernels, replaces hipcub::DeviceRadixSort) ------
// torchvision's nms orders candidates with scores.sort(stable, descending); the value sorted along is the candidate's
// index, so "stable descending" = ascending on the key ~asc(score) with ties in index order - exactly what an LSD radix
// sort with stable passes delivers.  Three passes of 11 / 11 / 10 bits over the n = count[0] live candidates (the
// count stays on the device: grids are sized for `cap`, tiles beyond n retire at once):
//   sort_hist_kernel    per-tile digit histogram (LDS atomics: counts are order-free)      -> hist[digit][tile]
//   sort_scan_kernel    one workgroup: exclusive prefix over (digit-major, tile-minor)     -> hist becomes offsets
//   sort_scatter_kernel per tile, 256 elements per round IN INDEX ORDER: a lane's rank among equal digits = equal
//                       digits of earlier rounds (run[d]) + of earlier waves this round (cnt[w][d]) + of earlier lanes of
//                       its wave (ballot match over the digit's bits) - no atomics decide an order, so every pass is
//                       stable and the result is a function of the input alone
constexpr int SORT_THREADS = 256, SORT_ROUNDS = 16, SORT_TILE = SORT_THREADS * SORT_ROUNDS, SORT_BINS = 2048;

__device__ __forceinline__ unsigned sort_key_desc(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  const unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending total order on the bits
  return ~asc;
}

struct SortPass {
  const float* score;      // pass 0: keys are derived from the scores and the value is the index itself
  const unsigned* key_in; const int* val_in;
  unsigned* key_out; int* val_out;
  int* hist;               // [SORT_BINS][tiles]
  const int* count;
  int tiles, shift, bits, first;
};

__device__ __forceinline__ unsigned sort_load_key(const SortPass& p, int i) {
  return p.first ? sort_key_desc(p.score[i]) : p.key_in[i];
}

__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(SortPass p) {
  __shared__ int h[SORT_BINS];
  const int n = p.count[0], tile = blockIdx.x, nb = 1 << p.bits;
  for (int d = threadIdx.x; d < nb; d += SORT_THREADS) h[d] = 0;
  __syncthreads();
  const int t0 = tile * SORT_TILE;
  if (t0 < n)
    for (int j = 0; j < SORT_ROUNDS; ++j) {
      const int i = t0 + j * SORT_THREADS + threadIdx.x;
      if (i < n) atomicAdd(&h[(sort_load_key(p, i) >> p.shift) & (nb - 1)], 1);
    }
  __syncthreads();
  for (int d = threadIdx.x; d < nb; d += SORT_THREADS) p.hist[(long)d * p.tiles + tile] = h[d];
}

// exclusive prefix of hist in (digit, tile) order; one workgroup of 1024 threads, two digits per thread at most
__global__ __launch_bounds__(1024) void sort_scan_kernel(SortPass p) {
  __shared__ int tot[SORT_BINS];
  __shared__ int wsum[16];
  const int nb = 1 << p.bits;
  for (int d = threadIdx.x; d < SORT_BINS; d += 1024) {
    int s = 0;
    if (d < nb)
      for (int t = 0; t < p.tiles; ++t) s += p.hist[(long)d * p.tiles + t];
    tot[d] = s;
  }
  __syncthreads();
  // block-wide exclusive scan of tot[0 .. 2048): thread t owns digits 2t, 2t+1
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int a = tot[2 * threadIdx.x], b = tot[2 * threadIdx.x + 1];
  int v = a + b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  if (lane == 63) wsum[w] = v;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < w; ++q) base += wsum[q];
  const int excl = base + v - (a + b);
  __syncthreads();
  tot[2 * threadIdx.x] = excl;
  tot[2 * threadIdx.x + 1] = excl + a;
  __syncthreads();
  for (int d = threadIdx.x; d < nb; d += 1024) {
    int run = tot[d];
    for (int t = 0; t < p.tiles; ++t) {
      const long k = (long)d * p.tiles + t;
      const int c = p.hist[k];
      p.hist[k] = run;
      run += c;
    }
  }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(SortPass p) {
  __shared__ int run[SORT_BINS];                     // this tile's next output slot per digit
  __shared__ int cnt[SORT_THREADS / 64][SORT_BINS];  // per wave and round; every entry is reset by whoever set it
  const int n = p.count[0], tile = blockIdx.x, nb = 1 << p.bits;
  const int t0 = tile * SORT_TILE;
  if (t0 >= n) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < nb; d += SORT_THREADS) {
    run[d] = p.hist[(long)d * p.tiles + tile];
#pragma unroll
    for (int q = 0; q < SORT_THREADS / 64; ++q) cnt[q][d] = 0;
  }
  __syncthreads();
  for (int j = 0; j < SORT_ROUNDS; ++j) {
    const int i = t0 + j * SORT_THREADS + threadIdx.x;
    const bool valid = i < n;
    unsigned key = 0;
    int val = 0, d = 0;
    if (valid) {
      key = sort_load_key(p, i);
      val = p.first ? i : p.val_in[i];
      d = (key >> p.shift) & (nb - 1);
    }
    // lanes of this wave with the same digit (and valid)
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < p.bits; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bal : ~bal;
    }
    const int lrank = __popcll(m & ((1ULL << lane) - 1ULL));
    const bool leader = valid && lrank == 0;
    if (leader) cnt[w][d] = __popcll(m);
    __syncthreads();
    if (valid) {
      int pos = run[d] + lrank;
      for (int q = 0; q < w; ++q) pos += cnt[q][d];
      p.key_out[pos] = key;
      p.val_out[pos] = val;
    }
    __syncthreads();
    if (leader) {
      atomicAdd(&run[d], cnt[w][d]);  // integer adds commute: run[d] is the same whatever order the waves arrive in
      cnt[w][d] = 0;
    }
    __syncthreads();
  }
}

struct NmsParams {
  const float* c_box; const float* c_score; const int* c_cls;
  const int* order;  // candidate ids sorted by descending score (stable)
  const int* count; const float* maxcoord;
  float thr; int topk; int per_class_from;
  int* keep;       // [topk] candidate ids
  int* n_keep;     // [1]
};

constexpr int NMS_MAXK = 1024;

// one wave per image
__global__ __launch_bounds__(64) void nms_topk_kernel(NmsParams p) {
  __shared__ float kb[NMS_MAXK][4];
  __shared__ float ka[NMS_MAXK];
  __shared__ int kc[NMS_MAXK];
  const int lane = threadIdx.x;
  const int n = p.count[0];
  // detectron2/layers/nms.py:19-29: >= 40000 boxes => per-class NMS on the raw boxes, no offsets
  const bool per_class = n >= p.per_class_from;
  const float offs = per_class ? 0.f : p.maxcoord[0] + 1.f;
  int nk = 0;
  for (int start = 0; start < n && nk < p.topk; start += 64) {
    const int i = start + lane;
    const bool valid = i < n;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0, area = 0;
    int id = -1, cls = -1;
    bool alive = valid;
    if (valid) {
      id = p.order[i];
      cls = p.c_cls[id];
      const float off = (float)cls * offs;  // idxs.to(boxes) * (boxes.max() + 1)
      x1 = p.c_box[4 * (long)id] + off; y1 = p.c_box[4 * (long)id + 1] + off;
      x2 = p.c_box[4 * (long)id + 2] + off; y2 = p.c_box[4 * (long)id + 3] + off;
      area = (x2 - x1) * (y2 - y1);
      for (int k = 0; k < nk && alive; ++k) {
        if (per_class && kc[k] != cls) continue;
        const float w = fmaxf(0.f, fminf(kb[k][2], x2) - fmaxf(kb[k][0], x1));
        const float h = fmaxf(0.f, fminf(kb[k][3], y2) - fmaxf(kb[k][1], y1));
        const float inter = w * h;
        if (inter / (ka[k] + area - inter) > p.thr) alive = false;
      }
    }
    // resolve the chunk in order: the lowest alive lane is kept and suppresses later lanes
    unsigned long long pending = __ballot(alive);
    while (pending && nk < p.topk) {
      const int l = __ffsll((long long)pending) - 1;
      const float bx1 = __shfl(x1, l, 64), by1 = __shfl(y1, l, 64), bx2 = __shfl(x2, l, 64), by2 = __shfl(y2, l, 64);
      const float ba = __shfl(area, l, 64);
      const int bid = __shfl(id, l, 64);
      const int bcls = __shfl(cls, l, 64);
      if (lane == 0) { kc[nk] = bcls; kb[nk][0] = bx1; kb[nk][1] = by1; kb[nk][2] = bx2; kb[nk][3] = by2; ka[nk] = ba; p.keep[nk] = bid; }
      ++nk;
      if (alive && lane > l && (!per_class || bcls == cls)) {
        const float w = fmaxf(0.f, fminf(bx2, x2) - fmaxf(bx1, x1));
        const float h = fmaxf(0.f, fminf(by2, y2) - fmaxf(by1, y1));
        const float inter = w * h;
        if (inter / (ba + area - inter) > p.thr) alive = false;
      }
      if (lane == l) alive = false;
      pending = __ballot(alive);
    }
    __syncthreads();
  }
  if (lane == 0) p.n_keep[0] = nk;
}

}  // namespace

namespace {

__global__ void gather_kernel(const float* c_box, const float* c_score, const int* c_row, const int* c_cls,
                              const int* keep, const int* n_keep, float* ob, float* os, int* oc, int* orow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_keep[0]) return;
  const int id = keep[i];
  for (int e = 0; e < 4; ++e) ob[4 * i + e] = c_box[4 * (long)id + e];
  os[i] = c_score[id]; oc[i] = c_cls[id]; orow[i] = c_row[id];
}

struct Ws {
  float* c_box; float* c_score; unsigned* key0; int* c_row; int* c_cls; int* val0; int* order; int* count;
  float* maxcoord; unsigned* key1; int* hist;
};

inline int sort_tiles(int cap) { return (cap + SORT_TILE - 1) / SORT_TILE; }
inline long ws_fixed_bytes(int cap) { return (long)cap * (16 + 4 * 7) + 256 + (long)SORT_BINS * sort_tiles(cap) * 4; }

inline Ws carve(void* workspace, long bytes, int cap) {
  (void)bytes;
  Ws k;
  char* w = (char*)workspace;
  k.c_box = (float*)w; w += (long)cap * 16;
  k.c_score = (float*)w; w += (long)cap * 4;
  k.key0 = (unsigned*)w; w += (long)cap * 4;   // sort ping (doubles as the finite-row map until pass 1 writes it)
  k.c_row = (int*)w; w += (long)cap * 4;
  k.c_cls = (int*)w; w += (long)cap * 4;
  k.val0 = (int*)w; w += (long)cap * 4;
  k.order = (int*)w; w += (long)cap * 4;        // sort pong values = the final order
  k.key1 = (unsigned*)w; w += (long)cap * 4;
  k.count = (int*)w; k.maxcoord = (float*)(w + 8); w += 256;
  k.hist = (int*)w;
  return k;
}

}  // namespace

// ---- test-time augmentation (wsl/modeling/test_time_augmentation_avg.py:269-294) --------------------------------
// One augmentation's predictions folded into the running averages on the device: every predicted box is mapped back
// to the original image - HFlipTransform.inverse (x -> W' - x) then ResizeTransform.inverse (x * sx, y * sy) applied
// to its four corners, then their bounding box, in float32 exactly like the numpy code the reference runs on the
// host - and added to acc_boxes; the scores are added to acc_scores.  The last augmentation divides by n_aug (torch.mean).
// The reference moves every augmentation's [R, 4K] boxes to the host and back for this.
__global__ void tta_accumulate_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                      float* __restrict__ acc_boxes, float* __restrict__ acc_scores, long nbox, long nscore,
                                      float sx, float sy, float flip_w, int first, int n_final) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < nbox) {
    const float* b = boxes + 4 * i;
    float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
    if (flip_w >= 0.f) { x0 = flip_w - x0; x1 = flip_w - x1; }
    x0 = x0 * sx; x1 = x1 * sx; y0 = y0 * sy; y1 = y1 * sy;
    const float o[4] = {fminf(x0, x1), fminf(y0, y1), fmaxf(x0, x1), fmaxf(y0, y1)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = first ? o[e] : acc_boxes[4 * i + e] + o[e];
      if (n_final > 0) v = v / (float)n_final;
      acc_boxes[4 * i + e] = v;
    }
  }
  if (i < nscore) {
    float v = first ? scores[i] : acc_scores[i] + scores[i];
    if (n_final > 0) v = v / (float)n_final;
    acc_scores[i] = v;
  }
}

extern "C" {

// bytes of scratch drn_detect_topk needs for up to `cap` candidates (cap = R*K is always enough)
long drn_detect_workspace_bytes(int cap) { return ws_fixed_bytes(cap < 1 ? 1 : cap) + 256; }

// Single image.  boxes [R][4*nreg], scores [R][K+1] (last column = background).  Leaves the kept
// candidate ids (descending score) in keep_ids[0..n_keep) on the device; drn_detect_gather expands them.
int drn_detect_topk(const float* boxes, const float* scores, int R, int K, int nreg, float img_h, float img_w,
                    float score_thresh, float nms_thresh, int topk, void* workspace, long workspace_bytes, int cap,
                    int* keep_ids, int* n_keep, void* stream) {
  if (!boxes || !scores || !workspace || !keep_ids || !n_keep || topk < 1 || topk > NMS_MAXK || cap < 1 || R < 0)
    return DRN_ERR_ARG;
  if (nreg != 1 && nreg != K) return DRN_ERR_ARG;
  if (cap < R) return DRN_ERR_ARG;
  if (workspace_bytes < drn_detect_workspace_bytes(cap)) return DRN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  Ws k = carve(workspace, workspace_bytes, cap);
  CandParams cp{boxes, scores, R, K, nreg, img_h, img_w, score_thresh, k.c_box, k.c_score, k.c_row, k.c_cls, k.count,
                k.maxcoord, cap, (int*)k.key0};  // key0 doubles as the row map until the sort's second pass overwrites it
  if (R > 0) hipLaunchKernelGGL(rows_finite_kernel, dim3((R + 3) / 4), dim3(256), 0, st, cp);
  hipLaunchKernelGGL(rows_number_kernel, dim3(1), dim3(1024), 0, st, cp);
  const long total = (long)R * K;
  const int ntiles = (int)((total + CAND_TILE - 1) / CAND_TILE);
  if (ntiles > SORT_BINS) return DRN_ERR_UNSUPPORTED;  // (8M scores per image; the tile counts live in the sort's histogram area)
  if (ntiles > 0) {
    hipLaunchKernelGGL(candidates_kernel<0>, dim3(ntiles), dim3(CAND_THREADS), 0, st, cp, k.hist, ntiles);
    hipLaunchKernelGGL(candidates_kernel<1>, dim3(ntiles), dim3(CAND_THREADS), 0, st, cp, k.hist, ntiles);
  } else {
    hipMemsetAsync(k.count, 0, sizeof(int), st);
  }
  // stable descending order of the live candidates: (score, index) -> key1/order -> key0/val0 -> key1/order
  const int tiles = sort_tiles(cap);
  const int shifts[3] = {0, 11, 22}, bits[3] = {11, 11, 10};
  for (int ps = 0; ps < 3; ++ps) {
    const bool even = (ps & 1) == 0;
    SortPass sp{k.c_score, even ? k.key0 : k.key1, even ? k.val0 : k.order, even ? k.key1 : k.key0,
                even ? k.order : k.val0, k.hist, k.count, tiles, shifts[ps], bits[ps], ps == 0};
    hipLaunchKernelGGL(sort_hist_kernel, dim3(tiles), dim3(SORT_THREADS), 0, st, sp);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, st, sp);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(tiles), dim3(SORT_THREADS), 0, st, sp);
  }
  NmsParams np{k.c_box, k.c_score, k.c_cls, k.order, k.count, k.maxcoord, nms_thresh, topk, 40000, keep_ids, n_keep};
  hipLaunchKernelGGL(nms_topk_kernel, dim3(1), dim3(64), 0, st, np);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_detect_gather(const void* workspace, long workspace_bytes, int cap, const int* keep_ids, const int* n_keep,
                      int topk, float* out_boxes, float* out_scores, int* out_classes, int* out_rows, void* stream) {
  if (!workspace || !keep_ids || !n_keep || !out_boxes || !out_scores || !out_classes || !out_rows) return DRN_ERR_ARG;
  Ws k = carve((void*)workspace, workspace_bytes, cap);
  hipLaunchKernelGGL(gather_kernel, dim3((topk + 63) / 64), dim3(64), 0, (hipStream_t)stream, k.c_box, k.c_score,
                     k.c_row, k.c_cls, keep_ids, n_keep, out_boxes, out_scores, out_classes, out_rows);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_tta_accumulate(const float* boxes, const float* scores, float* acc_boxes, float* acc_scores, long n_boxes,
                       long n_scores, float sx, float sy, float flip_w, int first, int n_final, void* stream) {
  if (!boxes || !scores || !acc_boxes || !acc_scores || n_boxes < 0 || n_scores < 0) return DRN_ERR_ARG;
  const long n = n_boxes > n_scores ? n_boxes : n_scores;
  if (n == 0) return DRN_OK;
  hipLaunchKernelGGL(tta_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                     scores, acc_boxes, acc_scores, n_boxes, n_scores, sx, sy, flip_w, first, n_final);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

}  // extern "C"
