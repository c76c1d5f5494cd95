// Linear sum assignment of the matcher's cost matrices ON THE DEVICE -- the step the dependency runs on the host with
// scipy.optimize.linear_sum_assignment after a device-to-host copy of every cost matrix (HF:474).
//
// Why: the assignment is tiny (100 queries x 16 targets per image and prediction level) but it is the one point of a train step
// where the GPU waits for the host: copy 160 matrices down, 160 scipy calls (2-3 ms), build index tensors, copy them up -- 5-8 ms
// of a 185 ms step with nothing queued behind it.  Solved here, the loss's index tensors are built by device ops and the step
// has no host synchronisation left in the loss.
//
// Bit-exact by construction: this is scipy's own algorithm (the shortest augmenting path of Crouse, "On implementing 2D
// rectangular assignment algorithms", as written in scipy/optimize/rectangular_lsap/rectangular_lsap.cpp of scipy 1.15) with the
// same arithmetic (float64 on the float32 costs, the same evaluation order), the same scan order (`remaining` filled in reverse
// and compacted by swap-with-last), the same tie rule (among equal shortest-path costs a column without a row wins, the last
// such in scan order; otherwise the first) and the same output order (sorted by row; a matrix with fewer columns than rows is
// solved transposed).  A Python transcription of the same steps is held to scipy on thousands of matrices -- tie-heavy integer
// ones included -- in tests/test_host_cpu.py, and the kernel to scipy on the GPU.
//
// One wave per problem: the scan over the remaining columns (<= 1024) is spread over the 64 lanes; each lane applies the scan
// rule to its positions (lane, lane + 64, ...: ascending) and keeps (lowest, first position with it, last position with it whose
// column is free); the wave combines them: lowest = min; index = the largest "last free" among the lanes that hold the minimum
// if any, else the smallest "first" -- exactly what the sequential scan returns.
#include "common.h"

namespace wm2f {
namespace {

constexpr int kLsaMax = 1024;  // larger side of a problem

struct LsaArgs {
  const float* cost;      // [problems][Q][Tmax]
  const int32_t* counts;  // [B] targets per image
  int32_t* rows;          // [problems][Tcap]  matched query, ascending
  int32_t* cols;          // [problems][Tcap]  its target
  int B, Q, Tmax, Tcap;
};

__global__ __launch_bounds__(64) void lsa_kernel(LsaArgs a) {
  __shared__ double u[kLsaMax], v[kLsaMax], spc[kLsaMax];
  __shared__ int path[kLsaMax], row4col[kLsaMax], col4row[kLsaMax], remaining[kLsaMax];
  __shared__ unsigned char SR[kLsaMax], SC[kLsaMax];
  const int lane = threadIdx.x, prob = blockIdx.x;
  const int T = a.counts[prob % a.B];
  if (T <= 0) return;
  const float* c = a.cost + (size_t)prob * a.Q * a.Tmax;
  const bool transpose = T < a.Q;               // scipy: fewer columns than rows -> solve the transposed problem
  const int nr = transpose ? T : a.Q, nc = transpose ? a.Q : T;
  // working matrix w(i, j): row i, column j of the (possibly transposed) problem
  auto w = [&](int i, int j) __attribute__((always_inline)) {
    return (double)(transpose ? c[(size_t)j * a.Tmax + i] : c[(size_t)i * a.Tmax + j]);
  };
  for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  __syncthreads();
  const double INF = __builtin_inf();

  for (int cur = 0; cur < nr; ++cur) {
    for (int j = lane; j < nc; j += 64) { remaining[j] = nc - j - 1; SC[j] = 0; spc[j] = INF; }
    for (int i = lane; i < nr; i += 64) SR[i] = 0;
    __syncthreads();
    double minVal = 0.0;
    int i = cur, num_remaining = nc, sink = -1;
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      double bval = INF;
      int first = 0x7fffffff, last_free = -1;
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = remaining[it];
        const double r = ((minVal + w(i, j)) - ui) - v[j];
        double s = spc[j];
        if (r < s) { path[j] = i; spc[j] = r; s = r; }
        const bool free_col = row4col[j] == -1;
        if (s < bval) { bval = s; first = it; last_free = free_col ? it : -1; }
        else if (s == bval && free_col) last_free = it;
      }
      // wave: lowest value, then the rule over the lanes that hold it
      double m = bval;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(m, d, 64);
        m = o < m ? o : m;
      }
      int f = bval == m ? first : 0x7fffffff, lf = bval == m ? last_free : -1;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const int of = __shfl_xor(f, d, 64), olf = __shfl_xor(lf, d, 64);
        f = of < f ? of : f;
        lf = olf > lf ? olf : lf;
      }
      const int index = lf >= 0 ? lf : f;
      minVal = m;
      // (scipy returns "infeasible" for an infinite minimum; the matcher's costs are clamped to +-1e10 and NaN-free)
      __syncthreads();  // every lane's spc / path updates are in LDS before anybody reads remaining[index] / row4col
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j;
      else i = row4col[j];
      --num_remaining;
      __syncthreads();
      if (lane == 0) {
        SC[j] = 1;
        remaining[index] = remaining[num_remaining];
      }
      __syncthreads();
    }
    // dual variables
    if (lane == 0) u[cur] += minVal;
    for (int i2 = lane; i2 < nr; i2 += 64)
      if (SR[i2] && i2 != cur) u[i2] += minVal - spc[col4row[i2]];
    for (int j2 = lane; j2 < nc; j2 += 64)
      if (SC[j2]) v[j2] -= minVal - spc[j2];
    __syncthreads();
    // augment the previous solution (a short chain: one lane)
    if (lane == 0) {
      int j = sink;
      while (true) {
        const int i2 = path[j];
        row4col[j] = i2;
        const int t = col4row[i2];
        col4row[i2] = j;
        j = t;
        if (i2 == cur) break;
      }
    }
    __syncthreads();
  }

  // output, sorted by row of the ORIGINAL matrix (query), as scipy returns it
  int32_t* ro = a.rows + (size_t)prob * a.Tcap;
  int32_t* co = a.cols + (size_t)prob * a.Tcap;
  if (!transpose) {  // rows 0 .. Q-1 in order, each with its target
    for (int i = lane; i < nr; i += 64) { ro[i] = i; co[i] = col4row[i]; }
  } else {           // target t has query col4row[t] (all distinct): rank by counting
    for (int t = lane; t < nr; t += 64) {
      const int qv = col4row[t];
      int rank = 0;
      for (int k = 0; k < nr; ++k) rank += col4row[k] < qv ? 1 : 0;
      ro[rank] = qv;
      co[rank] = t;
    }
  }
}

}  // namespace
}  // namespace wm2f

using namespace wm2f;

extern "C" int wm2f_lsa_batched(const void* cost, const void* counts, void* rows, void* cols, int problems, int B, int Q, int Tmax,
                                int Tcap, void* stream) {
  const char* who = "wm2f_lsa_batched";
  WM2F_REQUIRE(cost && counts && rows && cols, "%s: null pointer", who);
  WM2F_REQUIRE(problems > 0 && B > 0 && problems % B == 0 && Q > 0 && Tmax > 0 && Tcap > 0, "%s: sizes", who);
  if (Q > kLsaMax || Tmax > kLsaMax) {
    set_error("%s: a side of the cost matrix exceeds %d (solve on the host)", who, kLsaMax);
    return WM2F_EUNSUPPORTED;
  }
  LsaArgs a;
  a.cost = (const float*)cost;
  a.counts = (const int32_t*)counts;
  a.rows = (int32_t*)rows;
  a.cols = (int32_t*)cols;
  a.B = B;
  a.Q = Q;
  a.Tmax = Tmax;
  a.Tcap = Tcap;
  hipLaunchKernelGGL(lsa_kernel, dim3(problems), dim3(64), 0, (hipStream_t)stream, a);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
