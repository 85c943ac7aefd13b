// PCL refinement (proposal cluster learning) for gfx950: the targets the reference computes on the HOST with
// numpy + scikit-learn after a device->host copy of the scores, and the loss it computes on the HOST in C++
// (pcl_loss.h:52-131 always takes the CPU branch), as three device launches without a host round trip.
// Latency-bound integer / fp32 work, deterministic (fixed-order reductions, no float atomics).  Built with
// -ffp-contract=off: the float64 objective of the 1-D k-means and every IoU must round exactly like the oracle's.
//
// Replaces (reference file:line, paths under projects/WSL/wsl):
//   pcl_adjacency   _build_graph                        modeling/roi_heads/third_party/pcl.py:78-87
//                   (pairwise_iou: detectron2/structures/boxes.py:329-361)
//   pcl_refine      PCL(): _get_top_ranking_propoals, _get_graph_centers, _get_proposal_clusters
//                                                       modeling/roi_heads/third_party/pcl.py:26-200
//                   pcl_loss forward / backward          layers/csrc/pcl_loss/pcl_loss_cpu.cpp:8-117,
//                                                       layers/pcl_loss.py:10-93 (the / R scaling)
//                   PCLOutputs.pcl_loss + predict_probs  modeling/roi_heads/fast_rcnn.py:1725-1745, 1561-1575
//                   and the autograd of softmax under the custom backward
// Two steps of the reference are not functions of their inputs (scikit-learn's seeded k-means, numpy's unstable
// argsort on equal degrees); they follow the fixed definitions of oracle/pcl_oracle.py: the exact optimum of the 1-D
// 3-means objective, and "highest index among equal maxima".
#include "drn_common.h"
#include <float.h>

namespace {

constexpr int PCL_MAXR = 4096;          // proposals per image (BASELINE configs use 2000 / 4000)
constexpr int PCL_W = PCL_MAXR / 32;    // words of a row mask
constexpr int PCL_T = 1024;             // threads of the one workgroup that walks one refinement branch
constexpr int PCL_CHUNK = 64;           // blocked prefix sum (oracle/pcl_oracle.py SCAN_CHUNK)
constexpr int PCL_MAXP = 640;           // centres kept in LDS for the assignment phase (5 per labelled class)
constexpr int PCL_NINV = 3584;          // reciprocals kept in LDS for the k-means search (larger distances divide)
constexpr int PCL_ROWBUF_WORDS = (16384 + 32768 + 8192) / 4;  // adjacency rows of the top cluster staged in LDS

__device__ __forceinline__ float iou_xyxy(const float* a, const float* b) {
  const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
  const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
  const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
  const float ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  const float inter = iw * ih;
  return inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
}

// ------------------------------------------------------------------------------------------- adjacency bit matrix
struct PclAdjParams { const float* boxes; uint32_t* adj; int R; int W32; float thr; };

// block (256 rows) x word: one thread = 32 IoUs of its row against the word's 32 boxes (staged in LDS)
__global__ __launch_bounds__(256) void pcl_adjacency_kernel(PclAdjParams p) {
  __shared__ float cb[32][4];
  const int w = blockIdx.x;
  if (threadIdx.x < 128) {
    const int c = w * 32 + (threadIdx.x >> 2);
    cb[threadIdx.x >> 2][threadIdx.x & 3] = c < p.R ? p.boxes[4 * (long)c + (threadIdx.x & 3)] : 0.f;
  }
  __syncthreads();
  const int r = blockIdx.y * 256 + threadIdx.x;
  if (r >= p.R) return;
  float rb[4];
  for (int e = 0; e < 4; ++e) rb[e] = p.boxes[4 * (long)r + e];
  uint32_t bits = 0;
  for (int j = 0; j < 32; ++j)
    if (w * 32 + j < p.R && iou_xyxy(rb, cb[j]) > p.thr) bits |= 1u << j;
  p.adj[(long)r * p.W32 + w] = bits;
}

// ------------------------------------------------------------------------------------------- row softmax, all branches
struct PclSoftmaxParams { const float* logits; int ld; int cols[8]; int NB; int K1; int R; float* probs; };

__global__ __launch_bounds__(256) void pcl_softmax_kernel(PclSoftmaxParams p) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (r >= p.R) return;
  const float* x = p.logits + (long)r * p.ld + p.cols[b];
  float m = -FLT_MAX;
  for (int j = 0; j < p.K1; ++j) m = fmaxf(m, x[j]);
  float s = 0.f;
  for (int j = 0; j < p.K1; ++j) s += expf(x[j] - m);
  float* o = p.probs + ((long)b * p.R + r) * p.K1;
  for (int j = 0; j < p.K1; ++j) o[j] = expf(x[j] - m) / s;
}

// ------------------------------------------------------------------------------------------- one branch, one workgroup
struct PclRefineParams {
  const float* boxes;       // [R,4]
  const uint32_t* adj;      // [R,W32]
  const float* wsddn;       // [R,ld_ws]: the MIL head's scores (last_score of branch 0; class c in column c)
  int ld_ws;
  const float* probs;       // [NB][R][K1]: softmax of every branch, column 0 = background
  const float* onehot;      // [K] image-level labels
  int R, K, K1, W32, NB, PMAX;
  int* labels; float* cls_w; int* assign;                                   // [NB][R]
  int* pc_labels; float* pc_probs; int* pc_count; float* img_w; int* pc_rows; float* pc_scores;  // [NB][PMAX]
  int* n_pc;                // [NB]
  float* losses;            // [NB]
  float* dlogits; int ld; int cols[8];                                      // [R,ld], branch b at cols[b]
};

struct PclShared {  // fixed-size part of the LDS image (the big arrays follow, carved from dynamic LDS)
  uint32_t alive[PCL_W], member[PCL_W], live[PCL_W], inds[PCL_W];
  int nz[PCL_W];
  int wpre[PCL_W];  // rows set in the mask words before word w (of `alive` while gathering, of `member` afterwards)
  int nnz, cnt;
  unsigned long long red_u[16];
  double red_d[16];
  int red_i[16], red_j[16];
};

__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, PclShared& s) {
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  __syncthreads();  // protects red_u from the previous use
  if ((threadIdx.x & 63) == 0) s.red_u[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long m = s.red_u[0];
  for (int i = 1; i < PCL_T / 64; ++i) m = s.red_u[i] > m ? s.red_u[i] : m;
  return m;
}

__device__ __forceinline__ double block_sum_f64(double v, PclShared& s) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s.red_d[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = s.red_d[0];
  for (int i = 1; i < PCL_T / 64; ++i) m += s.red_d[i];
  return m;
}

// best (gain, i, j): larger gain wins, equal gains -> smaller i (then smaller j: same thread scans j ascending)
__device__ __forceinline__ void block_best_cut(double& g, int& i, int& j, PclShared& s) {
  for (int o = 32; o > 0; o >>= 1) {
    const double og = __shfl_xor(g, o, 64);
    const int oi = __shfl_xor(i, o, 64), oj = __shfl_xor(j, o, 64);
    if (og > g || (og == g && (oi < i || (oi == i && oj < j)))) { g = og; i = oi; j = oj; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s.red_d[threadIdx.x >> 6] = g; s.red_i[threadIdx.x >> 6] = i; s.red_j[threadIdx.x >> 6] = j; }
  __syncthreads();
  g = s.red_d[0]; i = s.red_i[0]; j = s.red_j[0];
  for (int q = 1; q < PCL_T / 64; ++q) {
    const double og = s.red_d[q];
    const int oi = s.red_i[q], oj = s.red_j[q];
    if (og > g || (og == g && (oi < i || (oi == i && oj < j)))) { g = og; i = oi; j = oj; }
  }
}

__device__ __forceinline__ bool bit(const uint32_t* m, int r) { return (m[r >> 5] >> (r & 31)) & 1u; }
// 1.0 / d, correctly rounded either way (the table holds the same IEEE quotients)
__device__ __forceinline__ double recip(const double* inv, int d) { return d <= PCL_NINV ? inv[d - 1] : 1.0 / (double)d; }

#ifdef PCL_PROFILE  // phase clocks (s_memtime) of branch 0, written over the tail of pc_scores: tools/pcl_bench.py reads them
#define PCL_TICK(k) { const long long now_ = clock64(); prof_[k] += now_ - last_; last_ = now_; }
#else
#define PCL_TICK(k)
#endif

__global__ __launch_bounds__(PCL_T) void pcl_refine_kernel(PclRefineParams p) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  PclShared& S = *reinterpret_cast<PclShared*>(lds_raw);
  unsigned char* big = lds_raw + ((sizeof(PclShared) + 15) & ~size_t(15));
  // k-means phase                                   | graph phase            | assignment phase
  float* sv = reinterpret_cast<float*>(big);         // sorted values [4096]  | centre boxes [PMAXP][4] + scores + labels
  double* P1 = reinterpret_cast<double*>(big + 16384);          // prefix [4097] -> Pc[j] | keep_row / keep_score
  double* T2 = reinterpret_cast<double*>(big + 16384 + 32776);  // chunk totals -> G0[j]  | deg [4096] int | w, assign
  double* B2 = reinterpret_cast<double*>(big + 16384 + 32776 + 32768);  // best 2-cluster gain of the prefix below cut j
  unsigned short* CP = reinterpret_cast<unsigned short*>(big + 16384 + 32776 + 65536);   // cut positions [4096]
  unsigned short* OPT = reinterpret_cast<unsigned short*>(big + 16384 + 32776 + 65536 + 8192);  // best lower cut of j
  double* INV = reinterpret_cast<double*>(big + 16384 + 32776 + 65536 + 16384);  // 1/d for d <= PCL_NINV (what LDS has left)
  // graph phase (the k-means arrays are dead by then)
  float* rval = sv;                                    // clipped score of every row [4096]
  unsigned short* keep_row = reinterpret_cast<unsigned short*>(P1);                                        // [4096]
  float* keep_score = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(P1) + 8192);                // [4096]
  unsigned short* rslot = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(P1) + 24576);  // row -> slot
  int* deg = reinterpret_cast<int*>(T2);               // degree of every member, by slot [4096]
  uint32_t* rowbuf = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(T2) + 16384);  // staged member rows:
                                                       // upper half of T2 + B2 + CP = 56 KB
  unsigned short* mlist = OPT;                         // members, ascending rows [4096]

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = p.R, K = p.K, K1 = p.K1, W32 = p.W32;
  const float* last = b == 0 ? p.wsddn : p.probs + (long)(b - 1) * R * K1 + 1;  // class c of row r: last[r*ldl + c]
  const int ldl = b == 0 ? p.ld_ws : K1;
  const float* pnew = p.probs + (long)b * R * K1;
  int* o_labels = p.labels + (long)b * R;
  float* o_w = p.cls_w + (long)b * R;
  int* o_assign = p.assign + (long)b * R;
  int* o_pcl = p.pc_labels + (long)b * p.PMAX;
  float* o_pcp = p.pc_probs + (long)b * p.PMAX;
  int* o_pcc = p.pc_count + (long)b * p.PMAX;
  float* o_iw = p.img_w + (long)b * p.PMAX;
  int* o_rows = p.pc_rows + (long)b * p.PMAX;
  float* o_sc = p.pc_scores + (long)b * p.PMAX;
  const float EPS9 = 1e-9f;  // pcl.py:33-37 (the upper clip 1 - 1e-9 rounds to 1.0f and never fires)

  if (tid < PCL_W) S.alive[tid] = tid < W32 ? (tid == W32 - 1 && (R & 31) ? (1u << (R & 31)) - 1u : 0xFFFFFFFFu) : 0u;
  __syncthreads();

#ifdef PCL_PROFILE
  long long prof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
  for (int d = tid; d < PCL_NINV; d += PCL_T) INV[d] = 1.0 / (double)(d + 1);
  __syncthreads();
  int P = 0;  // centres so far (uniform)
  for (int c = 0; c < K; ++c) {
    if (p.onehot[c] != 1.f) continue;
    // ---- pool = rows still alive; their clipped scores, compacted (order is irrelevant: they get sorted)
    int n = 0;
    for (int w = 0; w < W32; ++w) n += __popc(S.alive[w]);
    if (n == 0 || P + 5 > p.PMAX) continue;
    int npad = 64;
    while (npad < n) npad <<= 1;
    if (tid < PCL_W) {
      int acc = 0;
      for (int w = 0; w < tid; ++w) acc += __popc(S.alive[w]);
      S.wpre[tid] = acc;
    }
    __syncthreads();
    for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
      const int r = q * PCL_T + tid;
      if (r < R && bit(S.alive, r)) {
        const int pos = S.wpre[r >> 5] + __popc(S.alive[r >> 5] & ((1u << (r & 31)) - 1u));
        sv[pos] = fmaxf(last[(long)r * ldl + c], EPS9);
      }
    }
    for (int i = n + tid; i < npad; i += PCL_T) sv[i] = INFINITY;
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < npad; i += PCL_T) {
          const int x = i ^ j;
          if (x > i) {
            const float va = sv[i], vb = sv[x];
            const bool up = (i & k) == 0;
            if ((va > vb) == up) { sv[i] = vb; sv[x] = va; }
          }
        }
        // element i lives with thread i % 1024: partners at distance < 64 stay inside one wave, whose LDS accesses are
        // issued in program order - a workgroup barrier is only needed next to a pass that crosses waves
        const int nj = j > 1 ? (j >> 1) : k;  // distance of the next pass (first pass of the next stage: k)
        if (j >= 64 || nj >= 64) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    PCL_TICK(0)
    // ---- blocked prefix sums in float64 (sequential in chunks of 64, sequential over chunk totals)
    const int nch = (n + PCL_CHUNK - 1) / PCL_CHUNK;
    if (tid < nch) {
      double acc = 0.0;
      const int e = min(n, (tid + 1) * PCL_CHUNK);
      for (int i = tid * PCL_CHUNK; i < e; ++i) { acc += (double)sv[i]; P1[i + 1] = acc; }
      T2[tid] = acc;  // chunk total
    }
    if (tid == 0) P1[0] = 0.0;
    __syncthreads();
    if (tid < nch) {
      double off = 0.0;
      for (int u = 0; u < tid; ++u) off = off + T2[u];
      const int e = min(n, (tid + 1) * PCL_CHUNK);
      if (tid > 0)
        for (int i = tid * PCL_CHUNK; i < e; ++i) P1[i + 1] = off + P1[i + 1];
    }
    __syncthreads();
    // ---- cut positions (between distinct values), compacted in ascending order
    int ncut = 0;
    {
      // count per 1024-block with ballots; positions via wave prefix
      __shared__ int wcount[PCL_MAXR / 64 + 1];
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int i = q * PCL_T + tid;  // candidate cut position i in [1, n)
        const bool is = i >= 1 && i < n && sv[i - 1] < sv[i];
        const unsigned long long m = __ballot(is);
        if (lane == 0) wcount[q * 16 + wave] = __popcll(m);
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < PCL_MAXR / 64; ++i) { const int t = wcount[i]; wcount[i] = acc; acc += t; }
        wcount[PCL_MAXR / 64] = acc;
      }
      __syncthreads();
      ncut = wcount[PCL_MAXR / 64];
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int i = q * PCL_T + tid;
        const bool is = i >= 1 && i < n && sv[i - 1] < sv[i];
        const unsigned long long m = __ballot(is);
        if (is) CP[wcount[q * 16 + wave] + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
      }
      __syncthreads();
    }
    PCL_TICK(1)
    const int kk = min(3, min(n, ncut + 1));
    float thr = sv[0];
    if (kk >= 2) {
      // by cut index j: Pc[j] = P[cut j] (in P1), G0[j] = gain(0, cut j) (in T2); gain(cut j, n) is recomputed where used.
      // 1/m is an IEEE division wherever it appears (the oracle's 1.0 / m)
      const double pn = P1[n];
      double pc_[PCL_MAXR / PCL_T];
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int j = q * PCL_T + tid;
        if (j < ncut) pc_[q] = P1[CP[j]];
      }
      __syncthreads();
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int j = q * PCL_T + tid;
        if (j < ncut) { P1[j] = pc_[q]; T2[j] = pc_[q] * pc_[q] * (1.0 / (double)CP[j]); }
      }
      __syncthreads();
      double bg = -1.0;
      int bi = 0x7fffffff, bj = 0x7fffffff;
      if (kk == 2) {
        for (int j = tid; j < ncut; j += PCL_T) {
          const double d2 = pn - P1[j];
          const double g = T2[j] + d2 * d2 * (1.0 / (double)(n - CP[j]));
          if (g > bg) { bg = g; bi = j; bj = j; }
        }
      } else {
        // best lower cut of every upper cut j = 1..m by divide and conquer over the cut indices (the leftmost best lower
        // cut is non-decreasing in j: concave Monge cost): level h solves the odd multiples of h, bracketed by the
        // solved neighbours j-h and j+h; one wave per midpoint, lanes stride its candidates.  O(n log n) evaluations
        // instead of all pairs (which kept one CU busy for 0.28 ms per class at n = 2000).
        const int m = ncut - 1;
        int span = 1;
        while (span <= m) span <<= 1;
        for (int h = span >> 1; h >= 1; h >>= 1) {
          const int nmid = h <= m ? (m - h) / (2 * h) + 1 : 0;  // midpoints j = h (2q + 1) <= m
          // few midpoints with long candidate ranges: one wave each; many midpoints with short ranges: one thread each
          // (a midpoint whose range is long anyway goes to a small work list that the waves share afterwards)
          const bool per_thread = nmid >= 64;
          if (per_thread) {
            if (tid == 0) S.nnz = 0;
            __syncthreads();
            for (int q = tid; q < nmid; q += PCL_T) {
              const int j = h * (2 * q + 1);
              const int lo = (j - h >= 1) ? (int)OPT[j - h] : 0;
              int hi = (j + h <= m) ? (int)OPT[j + h] : j - 1;
              hi = max(min(hi, j - 1), lo);
              int slot = -1;
              if (hi - lo >= 48) slot = atomicAdd(&S.nnz, 1);
              if (slot >= 0 && slot < PCL_W) { S.nz[slot] = j; continue; }
              const double pj = P1[j];
              const int cj = CP[j];
              double bf = -1.0;
              int bx = lo;
              for (int i = lo; i <= hi; ++i) {
                const double d1 = pj - P1[i];
                const double f = T2[i] + d1 * d1 * recip(INV, cj - (int)CP[i]);
                if (f > bf) { bf = f; bx = i; }
              }
              OPT[j] = (unsigned short)bx;
              B2[j] = bf;
            }
            __syncthreads();
          }
          const int nwave_items = per_thread ? min(S.nnz, PCL_W) : nmid;
          for (int q = wave; q < nwave_items; q += PCL_T / 64) {
            const int j = per_thread ? S.nz[q] : h * (2 * q + 1);
            const int lo = (j - h >= 1) ? (int)OPT[j - h] : 0;
            int hi = (j + h <= m) ? (int)OPT[j + h] : j - 1;
            hi = max(min(hi, j - 1), lo);
            const double pj = P1[j];
            const int cj = CP[j];
            double bf = -1.0;
            int bx = 0x7fffffff;
            for (int i = lo + lane; i <= hi; i += 64) {
              const double d1 = pj - P1[i];
              const double f = T2[i] + d1 * d1 * recip(INV, cj - (int)CP[i]);
              if (f > bf) { bf = f; bx = i; }
            }
            for (int o = 32; o > 0; o >>= 1) {
              const double of = __shfl_xor(bf, o, 64);
              const int ox = __shfl_xor(bx, o, 64);
              if (of > bf || (of == bf && ox < bx)) { bf = of; bx = ox; }
            }
            if (lane == 0) { OPT[j] = (unsigned short)bx; B2[j] = bf; }
          }
          __syncthreads();
        }
        for (int j = 1 + tid; j <= m; j += PCL_T) {
          const double d2 = pn - P1[j];
          const double g = B2[j] + d2 * d2 * (1.0 / (double)(n - CP[j]));
          if (g > bg) { bg = g; bi = j; bj = j; }
        }
      }
      block_best_cut(bg, bi, bj, S);
      thr = sv[CP[bj]];
    }
    __syncthreads();
    PCL_TICK(2)
    // ---- members of the top cluster (row masks + a compact ascending list), their clipped scores by row, their
    // degrees inside the cluster; the members' adjacency rows are staged in LDS when they fit
    for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
      const int r = q * PCL_T + tid;
      const float v = r < R ? fmaxf(last[(long)r * ldl + c], EPS9) : 0.f;
      const bool mb = r < R && bit(S.alive, r) && v >= thr;
      const unsigned long long m = __ballot(mb);
      if (lane == 0) { S.member[(r >> 5)] = (uint32_t)m; S.member[(r >> 5) + 1] = (uint32_t)(m >> 32); }
      rval[r] = v;
    }
    __syncthreads();
    if (tid < PCL_W) {
      S.live[tid] = S.member[tid];
      int acc = 0;
      for (int w = 0; w < tid; ++w) acc += __popc(S.member[w]);
      S.wpre[tid] = acc;
    }
    int count = 0;
    for (int w = 0; w < W32; ++w) count += __popc(S.member[w]);
    const int nmem = count;
    const bool staged = nmem * W32 <= PCL_ROWBUF_WORDS;
    __syncthreads();
    for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
      const int r = q * PCL_T + tid;
      if (r < R && bit(S.member, r)) {
        const int x = S.wpre[r >> 5] + __popc(S.member[r >> 5] & ((1u << (r & 31)) - 1u));
        mlist[x] = (unsigned short)r;
        rslot[r] = (unsigned short)x;
        deg[x] = 0;
      }
    }
    __syncthreads();
    // one (member, word) pair per thread and step: independent coalesced loads; integer LDS adds are order-free
    for (int idx = tid; idx < nmem * W32; idx += PCL_T) {
      const int x = idx / W32, w = idx - x * W32;
      const uint32_t v = p.adj[(long)mlist[x] * W32 + w];
      if (staged) rowbuf[idx] = v;
      const int d = __popc(v & S.member[w]);
      if (d) atomicAdd(&deg[x], d);
    }
    __syncthreads();
    PCL_TICK(3)
    // ---- greedy graph centres (pcl.py:103-117) on ONE wave, no workgroup barriers: lane l owns the rows of mask words
    // l and l+64, so zeroing / decrementing degrees never crosses lanes.  Removing the set `inds` lowers the degree of
    // a live row r by |{j in inds : adj(r,j)}| = sum over j of adj[j][r] (IoU is symmetric): one row read per removed
    // node and a walk over its set bits, instead of every live row gathering over the removed set.
    if (wave == 0) {
      int nk = 0;
      while (true) {
        uint32_t key = 0;
        for (int x = lane; x < nmem; x += 64) {
          const uint32_t k2 = (((uint32_t)deg[x] << 12) | (uint32_t)x) + 1u;
          key = k2 > key ? k2 : key;
        }
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t2 = __shfl_xor(key, o, 64); key = t2 > key ? t2 : key; }
        const int tslot = (int)((key - 1u) & 4095u);
        const int t = mlist[tslot];
        const bool tlive = bit(S.live, t);
        const uint32_t* trow = staged ? rowbuf + tslot * W32 : p.adj + (long)t * W32;
        int cnt = 0;
        uint32_t iw[PCL_W / 64], lv[PCL_W / 64];
        uint32_t sk = 0;
        for (int h = 0; h < PCL_W / 64; ++h) {
          const int w = h * 64 + lane;
          lv[h] = S.live[w];
          iw[h] = (tlive && w < W32) ? (trow[w] & lv[h]) : 0u;
          cnt += __popc(iw[h]);
          lv[h] &= ~iw[h];
          S.live[w] = lv[h];  // this lane is the only reader / writer of its two words inside the loop
          uint32_t bits = iw[h];
          while (bits) {
            const int r = w * 32 + __ffs(bits) - 1;
            bits &= bits - 1u;
            const uint32_t v = __float_as_uint(rval[r]);
            sk = v > sk ? v : sk;
            deg[rslot[r]] = 0;
          }
        }
        for (int o = 32; o > 0; o >>= 1) {
          cnt += __shfl_xor(cnt, o, 64);
          const uint32_t t2 = __shfl_xor(sk, o, 64);
          sk = t2 > sk ? t2 : sk;
        }
        // the removed nodes j, listed in S.nz (128 per pass; more than one pass only for cliques above 128), then their
        // adjacency rows fetched eight at a time so that the row reads of one pass share their latency
        if (cnt) {
          int mine = 0;
          for (int h = 0; h < PCL_W / 64; ++h) mine += __popc(iw[h]);
          int incl = mine;
          for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
          }
          const int base = incl - mine;
          for (int pass = 0; pass * PCL_W < cnt; ++pass) {
            int gi = base;
            for (int h = 0; h < PCL_W / 64; ++h) {
              uint32_t bits = iw[h];
              while (bits) {
                const int j = (h * 64 + lane) * 32 + __ffs(bits) - 1;
                bits &= bits - 1u;
                if (gi / PCL_W == pass) S.nz[gi % PCL_W] = j;
                ++gi;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int nlist = min(cnt - pass * PCL_W, PCL_W);
            for (int g0 = 0; g0 < nlist; g0 += 8) {
              uint32_t jw[8][PCL_W / 64];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int j = S.nz[min(g0 + u, nlist - 1)];
                const uint32_t* jrow = staged ? rowbuf + (int)rslot[j] * W32 : p.adj + (long)j * W32;
#pragma unroll
                for (int g = 0; g < PCL_W / 64; ++g) {
                  const int w = g * 64 + lane;
                  jw[u][g] = (g0 + u < nlist && w < W32) ? jrow[w] : 0u;
                }
              }
#pragma unroll
              for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int g = 0; g < PCL_W / 64; ++g) {
                  uint32_t bits = jw[u][g] & lv[g];
                  while (bits) {
                    const int r = (g * 64 + lane) * 32 + __ffs(bits) - 1;
                    bits &= bits - 1u;
                    atomicSub(&deg[rslot[r]], 1);
                  }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          }
        }
        if (lane == 0) { keep_row[nk] = (unsigned short)t; keep_score[nk] = cnt ? __uint_as_float(sk) : 0.f; }
        ++nk;
        count -= cnt;
        __threadfence_block();
        if (count <= 5 || cnt == 0) break;
      }
      if (lane == 0) { S.nnz = nk; S.cnt = count; }
    }
    __syncthreads();
    const int nkeep = S.nnz;
    count = S.cnt;
    PCL_TICK(4)
#ifdef PCL_PROFILE
    prof_[8] += nkeep; prof_[9] += count;
#endif
    // ---- the min(nkeep, 5) best-scoring centres, descending, ties: later entry first (pcl.py:123-125)
    const int take = min(nkeep, 5);
    for (int s = 0; s < take; ++s) {
      unsigned long long key = 0;
      for (int i = tid; i < nkeep; i += PCL_T) {
        const float sc = keep_score[i];
        if (sc >= 0.f) {
          const unsigned long long k2 = (((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)i) + 1ull;
          key = k2 > key ? k2 : key;
        }
      }
      key = block_max_u64(key, S);
      const int i = (int)((key - 1ull) & 0xFFFFFFFFull);
      if (tid == 0) {
        const int row = keep_row[i];
        o_rows[P + s] = row;
        o_pcl[P + s] = c + 1;
        o_sc[P + s] = keep_score[i];
        keep_score[i] = -1.f;
        S.alive[row >> 5] &= ~(1u << (row & 31));  // np.delete of the chosen centre (pcl.py:134-135)
      }
      __syncthreads();
    }
    P += take;
    __syncthreads();
    PCL_TICK(5)
  }

  // ---- proposal clusters (pcl.py:143-200)
  __threadfence_block();
  __syncthreads();
  float* cb = sv;                                   // [PCL_MAXP][4]
  float* cs = sv + PCL_MAXP * 4;                    // [PCL_MAXP]
  int* cl = reinterpret_cast<int*>(cs + PCL_MAXP);  // [PCL_MAXP]   (4096 floats of sv hold 640*6 = 3840)
  float* roww = reinterpret_cast<float*>(T2);       // [4096] loss weight of each row
  short* rowa = reinterpret_cast<short*>(reinterpret_cast<unsigned char*>(T2) + 16384);  // [4096] assignment
  float* rowp = reinterpret_cast<float*>(B2);       // [4096] clipped probability of the assigned centre's class
  for (int i = tid; i < P; i += PCL_T) {
    const int row = o_rows[i];
    for (int e = 0; e < 4; ++e) cb[4 * i + e] = p.boxes[4 * (long)row + e];
    cs[i] = o_sc[i];
    cl[i] = o_pcl[i];
  }
  __syncthreads();
  double lsum = 0.0;  // this thread's share of the loss
  const float invR = 1.f / (float)R;
  for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
    const int r = q * PCL_T + tid;
    if (r >= R) continue;
    float rb[4];
    for (int e = 0; e < 4; ++e) rb[e] = p.boxes[4 * (long)r + e];
    float best = -1.f;
    int a = 0;
    for (int i = 0; i < P; ++i) {
      const float v = iou_xyxy(rb, cb + 4 * i);
      if (v > best) { best = v; a = i; }
    }
    int lab = 0, as = -1;
    float w = 0.f;
    if (P > 0) {
      w = best < 0.1f ? 0.f : cs[a];
      if (!(best < 0.5f)) { lab = cl[a]; as = a; }
    }
    o_labels[r] = lab; o_w[r] = w; o_assign[r] = as;
    roww[r] = w; rowa[r] = (short)as;
    rowp[r] = as >= 0 ? fmaxf(pnew[(long)r * K1 + lab], EPS9) : 0.f;
  }
  __syncthreads();
  // per centre: sum of weights, member count, mean clipped probability of its class (one wave per centre)
  for (int i = wave; i < P; i += PCL_T / 64) {
    double sw = 0.0, sp = 0.0;
    int n = 0;
    for (int r = lane; r < R; r += 64)
      if (rowa[r] == i) { sw += (double)roww[r]; sp += (double)rowp[r]; ++n; }
    for (int o = 32; o > 0; o >>= 1) { sw += __shfl_xor(sw, o, 64); sp += __shfl_xor(sp, o, 64); n += __shfl_xor(n, o, 64); }
    if (lane == 0) {
      o_iw[i] = (float)sw;
      o_pcc[i] = n;
      o_pcp[i] = (float)(sp / (double)n);  // 0/0 -> NaN like np.average of an empty selection
      cs[i] = (float)sw;                   // reuse: img_cls_loss_weights
      cb[4 * i] = (float)(sp / (double)n); // reuse: pc_probs
      cb[4 * i + 1] = (float)n;            // reuse: pc_count
    }
  }
  if (tid == 0) p.n_pc[b] = P;
  __syncthreads();
  PCL_TICK(6)
  // ---- loss (pcl_loss_cpu.cpp:8-58) and d loss / d logits through the softmax (pcl_loss_cpu.cpp:60-115)
  const int col0 = p.cols[b];
  float* rowg = rowp;          // d loss / d prob of the row's own column (the only non-zero entry of its row) / R
  float* rowgp = rowp + 4096;  // rowg * prob of that column
  for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
    const int r = q * PCL_T + tid;
    if (r >= R) continue;
    const int as = rowa[r];
    const int lab = as < 0 ? 0 : cl[as];
    const float pl = pnew[(long)r * K1 + lab];
    float g;
    if (as < 0) {
      lsum -= (double)roww[r] * (double)logf(fmaxf(pl, 1e-6f));
      g = -roww[r] / fmaxf(pl, 1e-5f);
    } else {
      g = p.onehot[lab - 1] != 0.f ? -cs[as] / fmaxf(cb[4 * as + 1] * cb[4 * as], 1e-5f) : 0.f;
    }
    g = g * invR;
    rowg[r] = g;
    rowgp[r] = g * pl;
    rowa[r] = (short)lab;
  }
  __syncthreads();
  // softmax backward of the one-hot row gradient, flattened over (row, column): coalesced, 8 loads in flight per thread
  const int total = R * K1;
  for (int base = tid; base < total; base += PCL_T * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * PCL_T;
      v[u] = idx < total ? pnew[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * PCL_T;
      if (idx < total) {
        const int r = idx / K1, j = idx - r * K1;
        p.dlogits[(long)r * p.ld + col0 + j] = v[u] * ((j == (int)rowa[r] ? rowg[r] : 0.f) - rowgp[r]);
      }
    }
  }
  for (int i = tid; i < P; i += PCL_T)
    if (p.onehot[cl[i] - 1] != 0.f) lsum -= (double)cs[i] * (double)logf(fmaxf(cb[4 * i], 1e-6f));
  const double tot = block_sum_f64(lsum, S);
  if (tid == 0) p.losses[b] = (float)(tot / (double)R);
  PCL_TICK(7)
#ifdef PCL_PROFILE
  if (tid == 0 && b == p.NB - 1)
    for (int k = 0; k < 10; ++k) o_sc[p.PMAX - 10 + k] = (float)prof_[k];
#endif
}

}  // namespace

extern "C" {

int drn_pcl_adjacency(const float* boxes, int R, float iou_thr, uint32_t* adj, hipStream_t stream) {
  if (!boxes || !adj || R <= 0) return DRN_ERR_ARG;
  if (R > PCL_MAXR) return DRN_ERR_UNSUPPORTED;
  PclAdjParams p{boxes, adj, R, (R + 31) / 32, iou_thr};
  pcl_adjacency_kernel<<<dim3(p.W32, (R + 255) / 256), 256, 0, stream>>>(p);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

int drn_pcl_refine(const float* logits, int ld, const int* cols, int n_branch, int K, const float* wsddn_scores,
                   int ld_ws, const float* boxes, const uint32_t* adj, const float* onehot, int R, float* probs,
                   int* labels, float* cls_w, int* assign, int* pc_labels, float* pc_probs, int* pc_count,
                   float* img_w, int* pc_rows, float* pc_scores, int* n_pc, int pmax, float* losses, float* dlogits,
                   hipStream_t stream) {
  if (!logits || !cols || !wsddn_scores || !boxes || !adj || !onehot || !probs || !labels || !cls_w || !assign ||
      !pc_labels || !pc_probs || !pc_count || !img_w || !pc_rows || !pc_scores || !n_pc || !losses || !dlogits)
    return DRN_ERR_ARG;
  if (R <= 0 || K <= 0 || n_branch <= 0 || n_branch > 8 || pmax < 5) return DRN_ERR_ARG;
  if (R > PCL_MAXR || K > 128 || pmax > PCL_MAXP) return DRN_ERR_UNSUPPORTED;
  PclSoftmaxParams sp{};
  sp.logits = logits; sp.ld = ld; sp.NB = n_branch; sp.K1 = K + 1; sp.R = R; sp.probs = probs;
  PclRefineParams rp{};
  rp.boxes = boxes; rp.adj = adj; rp.wsddn = wsddn_scores; rp.ld_ws = ld_ws; rp.probs = probs; rp.onehot = onehot;
  rp.R = R; rp.K = K; rp.K1 = K + 1; rp.W32 = (R + 31) / 32; rp.NB = n_branch; rp.PMAX = pmax;
  rp.labels = labels; rp.cls_w = cls_w; rp.assign = assign; rp.pc_labels = pc_labels; rp.pc_probs = pc_probs;
  rp.pc_count = pc_count; rp.img_w = img_w; rp.pc_rows = pc_rows; rp.pc_scores = pc_scores; rp.n_pc = n_pc;
  rp.losses = losses; rp.dlogits = dlogits; rp.ld = ld;
  for (int b = 0; b < n_branch; ++b) { sp.cols[b] = cols[b]; rp.cols[b] = cols[b]; }
  pcl_softmax_kernel<<<dim3((R + 255) / 256, n_branch), 256, 0, stream>>>(sp);
  DRN_CHECK_LAUNCH();
  const size_t lds = ((sizeof(PclShared) + 15) & ~size_t(15)) + 16384 + 32776 + 32768 + 32768 + 8192 + 8192 + PCL_NINV * 8;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pcl_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return DRN_ERR_LAUNCH;
    attr_set = true;
  }
  pcl_refine_kernel<<<n_branch, PCL_T, lds, stream>>>(rp);
  DRN_CHECK_LAUNCH();
  return DRN_OK;
}

}  // extern "C"
