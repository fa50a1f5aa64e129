// Triangular solves with a few right-hand sides (m <= 16 per pass): the single-vector solves of the
// likelihood (L^-1 z, gpmp/core/likelihood.py:46) and of the mean-space algebra (L^-1 [z, P]).
// HBM-bound (every element of L is read once: 4 n^2 bytes), so it runs as one small kernel per 128-row
// diagonal block instead of going through 128 x 128 MFMA tiles:
//   * x_k (the solution rows of block k) is already stored in B when step k starts;
//   * workgroup b applies it to its 128 rows, one row per thread pair:  b_rows -= L[rows, k] x_k ;
//   * workgroup 0 owns the NEXT diagonal block's rows: after its update it computes
//     x_{k+1} = inv(L_{k+1,k+1}) b_{k+1} (128 x 128 mat-vec) and stores it over b_{k+1} in place
//     (no other workgroup of this launch reads those rows).
// Every thread walks a contiguous piece of one row of L (16-byte loads), x_k is broadcast from LDS;
// no cross-lane reductions on the critical path.
#include "common.h"
#include <cstdlib>
#include <cstdint>
#include <map>
#include <mutex>
#include <utility>

namespace gpmp {
namespace {

// y[i] (i < jb) = sum_l M(i, l) v[l] for the 128 x 128 block Dinv (row-major, ld NB); TRANS uses Dinv^T.
// 256 threads: thread t handles row i = t & 127 and the half (t >> 7) of the l range; halves meet in LDS.
template <int R, bool TRANS>
__device__ __forceinline__ void block_matvec(const double* __restrict__ dinv, const double (*v)[R], double (*y)[R],
                                             double (*part)[R], int t) {
  const int i = t & 127, half = t >> 7;
  double acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0;
  if (!TRANS) {
    const double* row = dinv + i * NB + half * 64;
#pragma unroll 8
    for (int l = 0; l < 64; l += 2) {
      const d2 m2 = *reinterpret_cast<const d2*>(row + l);
#pragma unroll
      for (int c = 0; c < R; ++c) acc[c] = fma(m2[0], v[half * 64 + l][c], fma(m2[1], v[half * 64 + l + 1][c], acc[c]));
    }
  } else {
#pragma unroll 8
    for (int l = 0; l < 64; ++l) {
      const double m = dinv[(half * 64 + l) * NB + i];   // coalesced across i
#pragma unroll
      for (int c = 0; c < R; ++c) acc[c] = fma(m, v[half * 64 + l][c], acc[c]);
    }
  }
  if (half == 1) {
#pragma unroll
    for (int c = 0; c < R; ++c) part[i][c] = acc[c];
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int c = 0; c < R; ++c) y[i][c] = acc[c] + part[i][c];
  }
  __syncthreads();
}

// INIT launch (grid 1): x_first = op(Dinv_first) b_first stored in place.
// STEP launch for block k: B[rows] -= op(L)[rows, k] x_k for the rows still to be solved; the workgroup that
// owns the next diagonal block then turns it into x_next in place.
template <int R, bool TRANS, bool INIT>
__global__ void __launch_bounds__(256) trsv_kernel(const double* __restrict__ L, long ldl, const double* __restrict__ dinv,
                                                   double* __restrict__ B, long ldb, int n, int k, int m) {
  __shared__ double xs[NB][R];
  __shared__ double ys[NB][R];
  __shared__ double part[NB][R];
  const int t = threadIdx.x;
  const int nblk = (n + NB - 1) / NB;
  const int k0 = k * NB;
  const int jb = (n - k0) < NB ? (n - k0) : NB;
  if (INIT) {
    for (int idx = t; idx < NB * R; idx += 256) {
      const int l = idx / R, c = idx % R;
      xs[l][c] = (l < jb && c < m) ? B[(long)(k0 + l) * ldb + c] : 0.0;
    }
    __syncthreads();
    block_matvec<R, TRANS>(dinv + (size_t)k * NB * NB, xs, ys, part, t);
    for (int idx = t; idx < jb * R; idx += 256) {
      const int l = idx / R, c = idx % R;
      if (c < m) B[(long)(k0 + l) * ldb + c] = ys[l][c];
    }
    return;
  }
  // x_k from B
  for (int idx = t; idx < NB * R; idx += 256) {
    const int l = idx / R, c = idx % R;
    xs[l][c] = (l < jb && c < m) ? B[(long)(k0 + l) * ldb + c] : 0.0;
  }
  __syncthreads();
  // target block of this workgroup: forward -> blocks k+1+b ; backward -> blocks k-1-b
  const int tb = TRANS ? (k - 1 - (int)blockIdx.x) : (k + 1 + (int)blockIdx.x);
  const int r0 = tb * NB;
  const int rb = (n - r0) < NB ? (n - r0) : NB;
  const int i = t & 127, half = t >> 7;
  double acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0;
  if (i < rb) {
    if (!TRANS) {
      // row r0+i of L, columns k0 + half*64 .. +64 (contiguous per thread)
      const double* row = L + (long)(r0 + i) * ldl + k0 + half * 64;
      const int lmax = (jb - half * 64) < 64 ? (jb - half * 64) : 64;
      if (lmax == 64 && ((ldl & 1) == 0) && ((reinterpret_cast<uintptr_t>(L) & 15) == 0)) {
#pragma unroll 8
        for (int l = 0; l < 64; l += 2) {
          const d2 a2 = *reinterpret_cast<const d2*>(row + l);
#pragma unroll
          for (int c = 0; c < R; ++c) acc[c] = fma(a2[0], xs[half * 64 + l][c], fma(a2[1], xs[half * 64 + l + 1][c], acc[c]));
        }
      } else {
        for (int l = 0; l < lmax; ++l) {
          const double a0 = row[l];
#pragma unroll
          for (int c = 0; c < R; ++c) acc[c] = fma(a0, xs[half * 64 + l][c], acc[c]);
        }
      }
    } else {
      // (L^T)[r0+i, k0+l] = L[k0+l][r0+i]: coalesced across i
      const int lmax = (jb - half * 64) < 64 ? (jb - half * 64) : 64;
      for (int l = 0; l < lmax; ++l) {
        const double a = L[(long)(k0 + half * 64 + l) * ldl + r0 + i];
#pragma unroll
        for (int c = 0; c < R; ++c) acc[c] = fma(a, xs[half * 64 + l][c], acc[c]);
      }
    }
  }
  if (half == 1) {
#pragma unroll
    for (int c = 0; c < R; ++c) part[i][c] = acc[c];
  }
  __syncthreads();
  const bool owner_of_next = (blockIdx.x == 0);
  if (half == 0 && i < rb) {
#pragma unroll
    for (int c = 0; c < R; ++c) {
      if (c < m) {
        const double v = B[(long)(r0 + i) * ldb + c] - (acc[c] + part[i][c]);
        if (owner_of_next) xs[i][c] = v; else B[(long)(r0 + i) * ldb + c] = v;
      }
    }
  }
  if (!owner_of_next) return;
  // this workgroup holds the fully updated residual of the next diagonal block in xs: solve it
  if (half == 0 && i >= rb) {
#pragma unroll
    for (int c = 0; c < R; ++c) xs[i][c] = 0.0;
  }
  if (half == 0) {
#pragma unroll
    for (int c = 0; c < R; ++c) if (c >= m) xs[i][c] = 0.0;
  }
  __syncthreads();
  block_matvec<R, TRANS>(dinv + (size_t)tb * NB * NB, xs, ys, part, t);
  for (int idx = t; idx < rb * R; idx += 256) {
    const int l = idx / R, c = idx % R;
    if (c < m) B[(long)(r0 + l) * ldb + c] = ys[l][c];
  }
  (void)nblk;
}


// ---- one-launch variant: persistent workgroups, block hand-off through device memory ---------------------------
// The chain of launches above costs one kernel boundary (10-13 us) per 128-row block.  Here ONE launch of at most
// one workgroup per CU sweeps the whole triangle: workgroups draw row blocks in solve order from a ticket counter
// (a block only ever waits for blocks with smaller tickets, which have started: no deadlock whatever the dispatch
// order or residency), stream their 128 x 128 tiles of L as the x_k they need appear, and publish x_j for the later
// blocks.  Hand-off between workgroups (other CUs, other XCDs: per-CU L1 and per-XCD L2 are not coherent) follows the
// publish / consume recipe for gfx950: payload stored with agent-scope atomic stores (write-through), drained, ONE
// lane stores the flag with an agent-scope atomic; the consumer polls the flag with agent-scope atomic loads (bounded),
// then reads the payload with agent-scope atomic loads (they bypass the stale L1), so no acquire fence is needed.
// The tile of L that multiplies x_k is requested BEFORE the wait for x_k, and inv(L_jj) when the block starts, so the
// critical path per block is: flag -> 1 KB of x -> 2 small mat-vecs from registers -> store -> flag.
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define GPMP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int TRSV_MAXBLK = 4096;                 // n <= 524288
// state block: [0] ticket, [1] abort, [2..3] pad, [4 + k] flag of row block k.  Zeroed by hipMemsetAsync before a launch.
// One block per (device, stream) that has ever run a one-launch solve, allocated on first use and kept (16.4 KB each):
// solves on the same stream are ordered, solves on different streams never share a block, whatever their number in
// flight.  [1] doubles as the sticky "a solve on this stream gave up" word read by gpmp_solve_status.
constexpr size_t TRSV_STATE_WORDS = 4 + TRSV_MAXBLK;

struct TrsvP {
  const double* L;
  long ldl;
  const double* dinv;
  double* B;
  long ldb;
  int n, m, nblk;
  unsigned int* state;
  unsigned int spin_limit;      // polls of one wait before the solve gives up (GPMP_TRSV_SPIN_LOG2, default 2^26: tens of seconds)
};

template <int R, bool TRANS>
__global__ void __launch_bounds__(256) trsv_persist_kernel(TrsvP p) {
  __shared__ double xs[NB][R];
  __shared__ double part[NB][R];
  __shared__ int s_blk;
  gu32* st = (gu32*)p.state;
  const int t = threadIdx.x, i = t & 127, half = t >> 7;
  const bool vec_ok = ((p.ldl & 1) == 0) && ((reinterpret_cast<uintptr_t>(p.L) & 15) == 0);
  for (;;) {
    if (t == 0) s_blk = (int)__hip_atomic_fetch_add(st, 1u, GPMP_RLX_AGENT);
    __syncthreads();
    const int blk = s_blk;
    __syncthreads();                       // s_blk is overwritten in the next round
    if (blk >= p.nblk) return;
    const int j = TRANS ? p.nblk - 1 - blk : blk;
    const int j0 = j * NB;
    const int rb = (p.n - j0) < NB ? (p.n - j0) : NB;
    // inv(L_jj): this thread's half row (forward) / half column (transposed), held until the block's final mat-vec
    double dv[64];
    {
      const double* D = p.dinv + (size_t)j * NB * NB;
#pragma unroll
      for (int l = 0; l < 64; ++l) dv[l] = TRANS ? D[(half * 64 + l) * NB + i] : D[i * NB + half * 64 + l];
    }
    double acc[R];
#pragma unroll
    for (int c = 0; c < R; ++c) acc[c] = 0.0;
    const int nt = TRANS ? (p.nblk - 1 - j) : j;
    bool dead = false;
    for (int q = 0; q < nt; ++q) {
      const int k = TRANS ? p.nblk - 1 - q : q;
      const int k0 = k * NB;
      const int kb = (p.n - k0) < NB ? (p.n - k0) : NB;
      // this thread's 64 entries of the tile that multiplies x_k (requested before x_k is waited for)
      double a[64];
      const int lmax = (kb - half * 64) < 64 ? (kb - half * 64) : 64;
      if (i < rb) {
        if (!TRANS) {
          const double* row = p.L + (long)(j0 + i) * p.ldl + k0 + half * 64;
          if (lmax == 64 && vec_ok) {
#pragma unroll
            for (int l = 0; l < 64; l += 2) {
              const d2 v2 = *reinterpret_cast<const d2*>(row + l);
              a[l] = v2[0]; a[l + 1] = v2[1];
            }
          } else {
#pragma unroll
            for (int l = 0; l < 64; ++l) a[l] = (l < lmax) ? row[l] : 0.0;
          }
        } else {
          // (L^T)[j0 + i, k0 + l] = L[k0 + l][j0 + i]: coalesced across i
          const double* col = p.L + (long)(k0 + half * 64) * p.ldl + j0 + i;
#pragma unroll
          for (int l = 0; l < 64; ++l) a[l] = (l < lmax) ? col[(long)l * p.ldl] : 0.0;
        }
      } else {
#pragma unroll
        for (int l = 0; l < 64; ++l) a[l] = 0.0;
      }
      // wait for x_k: one lane polls (bounded), everybody else sits at the barrier
      if (t == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(st + 4 + k, GPMP_RLX_AGENT) == 0u) {
          if (__hip_atomic_load(st + 1, GPMP_RLX_AGENT) != 0u || ++spins > p.spin_limit) {
            __hip_atomic_store(st + 1, 1u, GPMP_RLX_AGENT);   // give up everywhere: the result is poisoned below
            s_blk = -1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      if (s_blk < 0) { dead = true; break; }
      // x_k (kb x m, zero padded) -> LDS with agent-scope atomic loads: they bypass this CU's L1, which may hold stale lines
      for (int idx = t; idx < NB * R; idx += 256) {
        const int l = idx / R, c = idx % R;
        double v = 0.0;
        if (l < kb && c < p.m) {
          const unsigned long long bits = __hip_atomic_load((gu64*)(p.B + (long)(k0 + l) * p.ldb + c), GPMP_RLX_AGENT);
          v = __builtin_bit_cast(double, bits);
        }
        xs[l][c] = v;
      }
      __syncthreads();
#pragma unroll
      for (int l = 0; l < 64; ++l)
#pragma unroll
        for (int c = 0; c < R; ++c) acc[c] = fma(a[l], xs[half * 64 + l][c], acc[c]);
      __syncthreads();                     // xs is rewritten by the next tile
    }
    if (dead) {
      // a producer never published: this block (and, through the abort word, every later one) gives up.  Its rows of
      // EVERY right-hand side become NaN, so nothing downstream can consume a half-solved vector silently; the abort
      // word stays set in the stream's state block until gpmp_solve_status reads it.
      if (half == 0 && i < rb)
        for (int c = 0; c < p.m; ++c) p.B[(long)(j0 + i) * p.ldb + c] = __builtin_nan("");
      return;
    }
    // b_j - sum: the two half-row partial sums meet in LDS; the block's own rows of B hold the right-hand side
    if (half == 1) {
#pragma unroll
      for (int c = 0; c < R; ++c) part[i][c] = acc[c];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int c = 0; c < R; ++c) {
        double v = 0.0;
        if (i < rb && c < p.m) v = p.B[(long)(j0 + i) * p.ldb + c] - (acc[c] + part[i][c]);
        xs[i][c] = v;
      }
    }
    __syncthreads();
    // x_j = op(inv(L_jj)) v from the registers loaded at the start
#pragma unroll
    for (int c = 0; c < R; ++c) acc[c] = 0.0;
#pragma unroll
    for (int l = 0; l < 64; ++l)
#pragma unroll
      for (int c = 0; c < R; ++c) acc[c] = fma(dv[l], xs[half * 64 + l][c], acc[c]);
    if (half == 1) {
#pragma unroll
      for (int c = 0; c < R; ++c) part[i][c] = acc[c];
    }
    __syncthreads();
    if (half == 0 && i < rb) {
#pragma unroll
      for (int c = 0; c < R; ++c)
        if (c < p.m) {
          const double v = acc[c] + part[i][c];
          __hip_atomic_store((gu64*)(p.B + (long)(j0 + i) * p.ldb + c), __builtin_bit_cast(unsigned long long, v), GPMP_RLX_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its payload stores
    __syncthreads();
    if (t == 0) __hip_atomic_store(st + 4 + j, 1u, GPMP_RLX_AGENT);
  }
}

struct StreamState { unsigned int* words; unsigned int* sticky; };
std::mutex g_state_mu;
std::map<std::pair<int, hipStream_t>, StreamState> g_states;

int state_for(hipStream_t st, StreamState& out, int& ncu) {
  int dev = 0;
  GPMP_HIP_TRY(hipGetDevice(&dev));
  auto it = g_states.find({dev, st});
  if (it == g_states.end()) {
    StreamState ns{nullptr, nullptr};
    // [state words | sticky abort counter]: the counter survives the per-launch memset of the state words
    GPMP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ns.words), sizeof(unsigned int) * (TRSV_STATE_WORDS + 4)));
    // zeroed ON THE OWNING STREAM: a hipMemset would run on the null stream, which a non-blocking stream (every PyTorch
    // side stream is one) does not wait for -- under load it could land in the middle of the first solve and reset the
    // ticket and the flags under the running kernel (seen as an occasional wrong or given-up first solve of a stream)
    GPMP_HIP_TRY(hipMemsetAsync(ns.words, 0, sizeof(unsigned int) * (TRSV_STATE_WORDS + 4), st));
    ns.sticky = ns.words + TRSV_STATE_WORDS;
    it = g_states.emplace(std::make_pair(dev, st), ns).first;
  }
  out = it->second;
  ncu = device_cu_count();
  return 0;
}

// abort word of the finished solve -> sticky counter of the stream (one thread; runs right behind the solve)
__global__ void trsv_latch_kernel(const unsigned int* state, unsigned int* sticky) {
  if (state[1] != 0u) sticky[0] += 1u;
}

template <int R>
int run_persist(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans, hipStream_t st) {
  const int nblk = (n + NB - 1) / NB;
  // memset + launch + latch must reach the stream as one unit: another host thread enqueueing a solve on the SAME stream
  // in between would run on this solve's dirty state
  std::lock_guard<std::mutex> lock(g_state_mu);
  StreamState ss;
  int ncu = 0;
  int rc = state_for(st, ss, ncu);
  if (rc) return rc;
  GPMP_HIP_TRY(hipMemsetAsync(ss.words, 0, sizeof(unsigned int) * (size_t)((4 + nblk + 3) / 4 * 4), st));
  static unsigned int spin_limit = 0;
  if (spin_limit == 0) {
    const char* e = getenv("GPMP_TRSV_SPIN_LOG2");
    const int lg = e ? atoi(e) : 26;
    spin_limit = 1u << (lg < 8 ? 8 : (lg > 31 ? 31 : lg));
  }
  TrsvP p{L, ldl, dinv, B, ldb, n, m, nblk, ss.words, spin_limit};
  const int grid = nblk < ncu ? nblk : ncu;
  if (!trans) hipLaunchKernelGGL((trsv_persist_kernel<R, false>), dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((trsv_persist_kernel<R, true>), dim3(grid), dim3(256), 0, st, p);
  GPMP_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(trsv_latch_kernel, dim3(1), dim3(1), 0, st, ss.words, ss.sticky);
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

template <int R>
int run(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans, hipStream_t st) {
  const int nblk = (n + NB - 1) / NB;
  if (!trans) {
    hipLaunchKernelGGL((trsv_kernel<R, false, true>), dim3(1), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, 0, m);
    for (int k = 0; k + 1 < nblk; ++k)
      hipLaunchKernelGGL((trsv_kernel<R, false, false>), dim3(nblk - 1 - k), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, k, m);
  } else {
    hipLaunchKernelGGL((trsv_kernel<R, true, true>), dim3(1), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, nblk - 1, m);
    for (int k = nblk - 1; k >= 1; --k)
      hipLaunchKernelGGL((trsv_kernel<R, true, false>), dim3(k), dim3(256), 0, st, L, ldl, dinv, B, ldb, n, k, m);
  }
  GPMP_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// In-place op(L)^-1 B for an n x m B with m <= TRSV_FEW_MAX (16): the single right-hand sides of the likelihood, and
// L^-1 [z, P] / L^-T (.) of the mean-space algebra with up to 15 mean columns (a linear mean in d = 8 has 9).  The sweep
// reads L once per pass of up to 8 columns; through the 128 x 128 MFMA tiles of the many-column solve, 10 columns cost as much
// as 128 (predict with a linear mean at n = 4096 / m = 10000: 9.3 -> 6.8 ms).
int trsv_few(const double* L, int n, long ldl, const double* dinv, double* B, int m, long ldb, int trans,
             hipStream_t st) {
  // one persistent launch from three blocks up to TRSV_MAXBLK; the launch-per-block chain outside that range
  const int nblk = (n + NB - 1) / NB;
  if (m > TRSV_FEW_MAX) { set_error("trsv_few: %d right-hand sides (at most %d)", m, TRSV_FEW_MAX); return -1; }
  // more than 8 columns: passes of 8 (and a last one of 1 / 2 / 4 / 8).  Each pass reads L again, but the sweep's inner loop
  // is LDS-bound beyond 8 columns (every fma fetches its x entry from LDS): measured at n = 4096 / 16384, forward:
  // 4 columns 0.21 / 0.93 ms, 8 columns 0.36 / 1.46 ms, a 16-column instantiation 0.96 / 3.9 ms.
  if (m > 8) {
    int rc = trsv_few(L, n, ldl, dinv, B, 8, ldb, trans, st);
    if (rc) return rc;
    return trsv_few(L, n, ldl, dinv, B + 8, m - 8, ldb, trans, st);
  }
  if (nblk >= 3 && nblk <= TRSV_MAXBLK) {
    if (m <= 1) return run_persist<1>(L, n, ldl, dinv, B, m, ldb, trans, st);
    if (m <= 2) return run_persist<2>(L, n, ldl, dinv, B, m, ldb, trans, st);
    if (m <= 4) return run_persist<4>(L, n, ldl, dinv, B, m, ldb, trans, st);
    return run_persist<8>(L, n, ldl, dinv, B, m, ldb, trans, st);
  }
  if (m <= 1) return run<1>(L, n, ldl, dinv, B, m, ldb, trans, st);
  if (m <= 2) return run<2>(L, n, ldl, dinv, B, m, ldb, trans, st);
  if (m <= 4) return run<4>(L, n, ldl, dinv, B, m, ldb, trans, st);
  return run<8>(L, n, ldl, dinv, B, m, ldb, trans, st);
}

// Number of single-vector solves on `stream` that gave up since the last call (0 in any healthy run); synchronises the
// stream.  status_host may be NULL (then only the return value is meaningful: 0 ok, > 0 count, < 0 error).
int solve_status(hipStream_t st, int* status_host) {
  std::lock_guard<std::mutex> lock(g_state_mu);
  int dev = 0;
  GPMP_HIP_TRY(hipGetDevice(&dev));
  auto it = g_states.find({dev, st});
  unsigned int v = 0;
  if (it != g_states.end()) {
    // everything on `st` itself: the null stream is not ordered with a non-blocking stream
    GPMP_HIP_TRY(hipMemcpyAsync(&v, it->second.sticky, sizeof(v), hipMemcpyDeviceToHost, st));
    GPMP_HIP_TRY(hipStreamSynchronize(st));
    if (v != 0) {
      GPMP_HIP_TRY(hipMemsetAsync(it->second.sticky, 0, sizeof(v), st));
      GPMP_HIP_TRY(hipStreamSynchronize(st));
    }
  }
  if (status_host) *status_host = (int)v;
  return (int)v;
}

// Forget `st`: wait for what is queued on it, free its state block, erase its map entries.  Without it the block (16.4 KB)
// lives until the process ends and a later stream that happens to get the same handle value inherits it (harmless: the
// state words are zeroed before every solve; only the give-up counter would carry over).  No HIP call when the stream
// never ran a single-vector solve.
int stream_release(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_state_mu);
  for (auto it = g_states.begin(); it != g_states.end();) {
    if (it->first.second != st) { ++it; continue; }
    GPMP_HIP_TRY(hipStreamSynchronize(st));       // a solve of this stream may still be polling the block
    GPMP_HIP_TRY(hipFree(it->second.words));
    it = g_states.erase(it);
  }
  return 0;
}

}  // namespace gpmp

extern "C" int gpmp_solve_status(gpmp_stream_t stream, int* status_host) {
  return gpmp::solve_status(gpmp::as_stream(stream), status_host);
}

extern "C" int gpmp_stream_release(gpmp_stream_t stream) { return gpmp::stream_release(gpmp::as_stream(stream)); }
