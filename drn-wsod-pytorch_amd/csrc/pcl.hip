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

__global__ __launch_bounds__(PCL_T) void pcl_refine_kernel(PclRefineParams p) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  PclShared& S = *reinterpret_cast<PclShared*>(lds_raw);
  unsigned char* big = lds_raw + ((sizeof(PclShared) + 15) & ~size_t(15));
  // k-means phase                                   | graph phase            | assignment phase
  float* sv = reinterpret_cast<float*>(big);         // sorted values [4096]  | centre boxes [PMAXP][4] + scores + labels
  double* P1 = reinterpret_cast<double*>(big + 16384);          // prefix [4097] -> Pc[j] | keep_row / keep_score
  double* T2 = reinterpret_cast<double*>(big + 16384 + 32776);  // term2 by cut [4096]    | deg [4096] int | w, assign
  double* INV = reinterpret_cast<double*>(big + 16384 + 32776 + 32768);  // 1/m [4096]
  unsigned short* CP = reinterpret_cast<unsigned short*>(big + 16384 + 32776 + 65536);  // cut positions [4096]
  int* deg = reinterpret_cast<int*>(T2);
  unsigned short* keep_row = reinterpret_cast<unsigned short*>(P1);
  float* keep_score = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(P1) + 8192);

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
  for (int m = tid; m < PCL_MAXR; m += PCL_T) INV[m] = 1.0 / (double)(m + 1);
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
    for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
      const int r = q * PCL_T + tid;
      const bool a = r < R && bit(S.alive, r);
      // position = alive rows before r
      int pos = 0;
      if (a) {
        for (int w = 0; w < (r >> 5); ++w) pos += __popc(S.alive[w]);
        pos += __popc(S.alive[r >> 5] & ((1u << (r & 31)) - 1u));
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
        __syncthreads();
      }
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
    const int kk = min(3, min(n, ncut + 1));
    float thr = sv[0];
    if (kk >= 2) {
      // by cut index j: Pc[j] = P[cut], T2[j] = gain(cut, n); P1 is re-read before being overwritten
      double pn = P1[n];
      double pc_[PCL_MAXR / PCL_T], t2_[PCL_MAXR / PCL_T];
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int j = q * PCL_T + tid;
        if (j < ncut) {
          const int cpos = CP[j];
          const double d2 = pn - P1[cpos];
          pc_[q] = P1[cpos];
          t2_[q] = d2 * d2 * INV[n - cpos - 1];
        }
      }
      __syncthreads();
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int j = q * PCL_T + tid;
        if (j < ncut) { P1[j] = pc_[q]; T2[j] = t2_[q]; }
      }
      __syncthreads();
      double bg = -1.0;
      int bi = 0x7fffffff, bj = 0x7fffffff;
      if (kk == 2) {
        for (int j = tid; j < ncut; j += PCL_T) {
          const double d = P1[j];
          const double g = d * d * INV[CP[j] - 1] + T2[j];
          if (g > bg) { bg = g; bi = j; bj = j; }
        }
      } else {
        for (int i = tid; i < ncut - 1; i += PCL_T) {
          const int c1 = CP[i];
          const double p1 = P1[i];
          const double g0 = p1 * p1 * INV[c1 - 1];
          for (int j = i + 1; j < ncut; ++j) {
            const double d1 = P1[j] - p1;
            const double g = (g0 + d1 * d1 * INV[CP[j] - c1 - 1]) + T2[j];
            if (g > bg) { bg = g; bi = i; bj = j; }
          }
        }
      }
      block_best_cut(bg, bi, bj, S);
      thr = sv[CP[bj]];
    }
    __syncthreads();
    // ---- members of the top cluster; degrees inside it
    for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
      const int r = q * PCL_T + tid;
      const bool mb = r < R && bit(S.alive, r) && fmaxf(last[(long)r * ldl + c], EPS9) >= thr;
      const unsigned long long m = __ballot(mb);
      if (lane == 0) { S.member[(r >> 5)] = (uint32_t)m; S.member[(r >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    if (tid < PCL_W) S.live[tid] = S.member[tid];
    int count = 0;
    for (int w = 0; w < W32; ++w) count += __popc(S.member[w]);
    for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
      const int r = q * PCL_T + tid;
      int d = 0;
      if (r < R && bit(S.member, r)) {
        const uint32_t* row = p.adj + (long)r * W32;
        for (int w = 0; w < W32; ++w) d += __popc(row[w] & S.member[w]);
      }
      deg[r] = d;
    }
    __syncthreads();
    // ---- greedy graph centres (pcl.py:103-117)
    int nkeep = 0;
    while (true) {
      unsigned long long key = 0;
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int r = q * PCL_T + tid;
        if (r < R && bit(S.member, r)) {
          const unsigned long long k2 = (((unsigned long long)deg[r] << 12) | (unsigned)r) + 1ull;
          key = k2 > key ? k2 : key;
        }
      }
      key = block_max_u64(key, S);
      const int t = (int)((key - 1ull) & 4095ull);
      const bool tlive = bit(S.live, t);
      if (tid < PCL_W) S.inds[tid] = (tlive && tid < W32) ? (p.adj[(long)t * W32 + tid] & S.live[tid]) : 0u;
      __syncthreads();
      if (wave == 0) {
        int cnt = 0, base = 0;
        for (int h = 0; h < PCL_W / 64; ++h) {
          const uint32_t v = S.inds[h * 64 + lane];
          cnt += __popc(v);
          const unsigned long long m = __ballot(v != 0u);
          if (v != 0u) S.nz[base + __popcll(m & ((1ull << lane) - 1ull))] = h * 64 + lane;
          base += __popcll(m);
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) { S.cnt = cnt; S.nnz = base; }
      }
      __syncthreads();
      const int cnt = S.cnt, nnz = S.nnz;
      unsigned long long sk = 0;
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int r = q * PCL_T + tid;
        if (r < R && bit(S.inds, r)) {
          const unsigned long long v = __float_as_uint(fmaxf(last[(long)r * ldl + c], EPS9));
          sk = v > sk ? v : sk;
        }
      }
      sk = block_max_u64(sk, S);
      if (tid < PCL_W) S.live[tid] &= ~S.inds[tid];
      __syncthreads();
      for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
        const int r = q * PCL_T + tid;
        if (r < R && bit(S.member, r)) {
          if (!bit(S.live, r)) deg[r] = 0;
          else if (nnz) {
            const uint32_t* row = p.adj + (long)r * W32;
            int d = 0;
            for (int z = 0; z < nnz; ++z) d += __popc(row[S.nz[z]] & S.inds[S.nz[z]]);
            deg[r] -= d;
          }
        }
      }
      if (tid == 0) { keep_row[nkeep] = (unsigned short)t; keep_score[nkeep] = cnt ? __uint_as_float((uint32_t)sk) : 0.f; }
      ++nkeep;
      count -= cnt;
      __syncthreads();
      if (count <= 5 || cnt == 0) break;
    }
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
  }

  // ---- proposal clusters (pcl.py:143-200)
  __threadfence_block();
  __syncthreads();
  float* cb = sv;                                   // [PCL_MAXP][4]
  float* cs = sv + PCL_MAXP * 4;                    // [PCL_MAXP]
  int* cl = reinterpret_cast<int*>(cs + PCL_MAXP);  // [PCL_MAXP]   (4096 floats of sv hold 640*6 = 3840)
  float* roww = reinterpret_cast<float*>(T2);       // [4096] loss weight of each row
  short* rowa = reinterpret_cast<short*>(reinterpret_cast<unsigned char*>(T2) + 16384);  // [4096] assignment
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
  }
  __syncthreads();
  // per centre: sum of weights, member count, mean clipped probability of its class (one wave per centre)
  for (int i = wave; i < P; i += PCL_T / 64) {
    double sw = 0.0, sp = 0.0;
    int n = 0;
    const int lbl = cl[i];
    for (int r = lane; r < R; r += 64)
      if (rowa[r] == i) { sw += (double)roww[r]; sp += (double)fmaxf(pnew[(long)r * K1 + lbl], EPS9); ++n; }
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
  // ---- loss (pcl_loss_cpu.cpp:8-58) and d loss / d logits through the softmax (pcl_loss_cpu.cpp:60-115)
  const int col0 = p.cols[b];
  for (int q = 0; q < PCL_MAXR / PCL_T; ++q) {
    const int r = q * PCL_T + tid;
    if (r >= R) continue;
    const float* pr = pnew + (long)r * K1;
    const int as = rowa[r];
    int lab = 0;
    float g = 0.f;
    if (as < 0) {
      const float p0 = pr[0];
      lsum -= (double)roww[r] * (double)logf(fmaxf(p0, 1e-6f));
      g = -roww[r] / fmaxf(p0, 1e-5f);
    } else {
      lab = cl[as];
      g = p.onehot[lab - 1] != 0.f ? -cs[as] / fmaxf(cb[4 * as + 1] * cb[4 * as], 1e-5f) : 0.f;
    }
    g = g * invR;
    const float gp = g * pr[lab];
    float* dl = p.dlogits + (long)r * p.ld + col0;
    for (int j = 0; j < K1; ++j) dl[j] = pr[j] * ((j == lab ? g : 0.f) - gp);
  }
  for (int i = tid; i < P; i += PCL_T)
    if (p.onehot[cl[i] - 1] != 0.f) lsum -= (double)cs[i] * (double)logf(fmaxf(cb[4 * i], 1e-6f));
  const double tot = block_sum_f64(lsum, S);
  if (tid == 0) p.losses[b] = (float)(tot / (double)R);
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
  const size_t lds = ((sizeof(PclShared) + 15) & ~size_t(15)) + 16384 + 32776 + 32768 + 32768 + 8192;
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
