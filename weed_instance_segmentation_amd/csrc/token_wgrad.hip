// Weight gradient of a token Linear:  dW[n][k] = sum_m dY[m][n] * X[m][k],  db[n] = sum_m dY[m][n]
// over the M = batch x tokens rows of the pixel decoder's encoder layers (HF:1036-1103: value_proj, sampling_offsets,
// attention_weights, output_proj, fc1, fc2 -- the backward of `nn.Linear` that autograd derives, TORCHF linear).
//
// Why a kernel: at config 2 (bf16 autocast, B = 16: M = 344 064) the library runs these 36 products as plain GEMMs with
// a 256 x 256 (or 1024 x 256) OUTPUT and a 344 k-long contraction -- 16 output tiles for 256 CUs: 547 us each, 33 ms per
// step (profiles/r03_train_bf16_b16_step_breakdown_start_of_round.txt).  The operation is HBM-bound: it reads the two
// (M, 256) bf16 operands once, 352 MB, i.e. ~75 us at the rate a streaming read sustains.
//
// Shape of the kernel (bound: HBM; MFMA has 3 x slack):
//   * split over the contraction: workgroup (split s, column block cb) owns tokens [s * span, (s + 1) * span) and a
//     256 x 256 block of dW entirely in registers (8 waves x (64 x 128) = 8 x 128 accumulator VGPRs), streams its
//     tokens in chunks of 64 through a double-buffered LDS image, and writes ONE fp32 partial tile to the workspace;
//     a second kernel adds the partials in split order -- deterministic, no atomics;
//   * both operands are row-major with the CONTRACTION index m as the row, but a bf16 MFMA wants its k index packed per
//     lane: the LDS image keeps the rows as they come from HBM (coalesced 16-byte pieces, staged through registers) and
//     the fragments are read with ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major);
//     16-byte pieces of row m are stored XOR-swizzled by (m & 3) << 2 so that the four rows of a block -- 512 bytes
//     apart, i.e. the same banks -- spread over all 64 banks;
//   * db comes from the same fragments: one more MFMA per dY fragment against a constant all-ones operand.
// Shapes: N, K multiples of 8; any M.  Rows / columns beyond M / N / K read as zeros through the buffer range check.
#include "common.h"

namespace wm2f {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

constexpr int kTile = 256;            // dW block per workgroup: 256 (n) x 256 (k)
constexpr int kMC = 64;               // tokens per chunk
constexpr int kThreadsW = 512;        // 8 waves: 4 (n) x 2 (k), each 64 x 128
constexpr int kRowBytes = kTile * 2;  // LDS row: 256 bf16
constexpr int kImgBytes = kMC * kRowBytes;                 // one operand, one chunk: 32 KiB
constexpr int kPieces = 2 * kImgBytes / 16 / kThreadsW;   // 16-byte pieces per thread and chunk: 8 (4 of dY, 4 of X)
constexpr unsigned kOobW = 0x80000000u;

struct WgradArgs {
  const void* dy;
  const void* x;
  float* ws_w;   // [cb][split][256][256]
  float* ws_b;   // [cbn][split][256]
  long long M;
  int N, K, cbn, cbk, splits, chunks_per_split;
};

// byte offset of the 16-byte piece `c` (0..31) of row `m` inside an operand image
__device__ __forceinline__ int img_off(int m, int c) { return m * kRowBytes + ((c ^ ((m & 3) << 2)) << 4); }

__global__ __launch_bounds__(kThreadsW) void token_wgrad_bf16_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2 buffers][dY image | X image]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, cb = blockIdx.y;
  const int cb_n = cb / a.cbk, cb_k = cb - cb_n * a.cbk;
  const long long m_begin = (long long)split * a.chunks_per_split * kMC;
  long long m_end = m_begin + (long long)a.chunks_per_split * kMC;
  if (m_end > a.M) m_end = a.M;
  const int n_chunks = m_begin < m_end ? (int)((m_end - m_begin + kMC - 1) / kMC) : 0;

  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)(a.M * a.N * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.M * a.K * 2), 0x00020000);

  // ---- staging: piece id = tid + 512 i; ids 0..2047 are dY (row = id >> 5, LDS position = id & 31), 2048.. are X
  int st_row[kPieces / 2], st_pos[kPieces / 2];
  unsigned st_coln[kPieces / 2], st_colk[kPieces / 2];  // byte offset of the piece inside a global row, or out of range
#pragma unroll
  for (int i = 0; i < kPieces / 2; ++i) {
    const int id = tid + kThreadsW * i;
    st_row[i] = id >> 5;
    st_pos[i] = id & 31;
    const int c = st_pos[i] ^ ((st_row[i] & 3) << 2);  // the global piece this LDS position holds
    const int coln = cb_n * kTile + c * 8, colk = cb_k * kTile + c * 8;
    st_coln[i] = coln < a.N ? (unsigned)(coln * 2) : kOobW;
    st_colk[i] = colk < a.K ? (unsigned)(colk * 2) : kOobW;
  }
  u32x4 stage[kPieces];
  auto issue_loads = [&](int chunk) __attribute__((always_inline)) {
    const long long m0 = m_begin + (long long)chunk * kMC;
#pragma unroll
    for (int i = 0; i < kPieces / 2; ++i) {
      const long long m = m0 + st_row[i];
      const bool ok = m < m_end;
      const unsigned on = (ok && st_coln[i] != kOobW) ? (unsigned)(m * a.N * 2) + st_coln[i] : kOobW;
      const unsigned ok_ = (ok && st_colk[i] != kOobW) ? (unsigned)(m * a.K * 2) + st_colk[i] : kOobW;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(dy_rs, (int)on, 0, 0);
      stage[kPieces / 2 + i] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)ok_, 0, 0);
    }
  };
  auto write_stage = [&](int buf) __attribute__((always_inline)) {
    unsigned char* base = lds + buf * 2 * kImgBytes;
#pragma unroll
    for (int i = 0; i < kPieces / 2; ++i) {
      const int off = st_row[i] * kRowBytes + st_pos[i] * 16;
      *reinterpret_cast<u32x4*>(base + off) = stage[i];
      *reinterpret_cast<u32x4*>(base + kImgBytes + off) = stage[kPieces / 2 + i];
    }
  };

  // ---- fragment addressing (ds_read_b64_tr_b16): lane = 16 g + 4 q + p supplies row q, columns 4 p .. 4 p + 3 of its
  // group's 4 x 16 block; group g covers MFMA rows 16 (g & 1) .. + 15 and contraction rows 8 (g >> 1) + q (+ 4)
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int wn = wave & 3, wk = wave >> 2;
  const int lane_row = 8 * (g >> 1) + q;
  int offA[2], offB[4];
#pragma unroll
  for (int t = 0; t < 2; ++t) offA[t] = img_off(lane_row, (64 * wn + 32 * t) / 8 + 2 * (g & 1) + (p >> 1)) + 8 * (p & 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) offB[t] = kImgBytes + img_off(lane_row, (128 * wk + 32 * t) / 8 + 2 * (g & 1) + (p >> 1)) + 8 * (p & 1);

  f32x16 acc[2][4], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    accb[i] = f32x16{};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{};
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

  auto frag = [&](const unsigned char* buf, int off, int ks) __attribute__((always_inline)) {
    // rows 16 ks + lane_row (elements 0..3) and + 4 (elements 4..7); (row & 3) == q in both, so the swizzle term in `off` holds
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(buf + off + (16 * ks) * kRowBytes));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(buf + off + (16 * ks + 4) * kRowBytes));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  if (n_chunks > 0) {
    issue_loads(0);
    write_stage(0);
    __syncthreads();
    if (n_chunks > 1) issue_loads(1);
    for (int c = 0; c < n_chunks; ++c) {
      const unsigned char* buf = lds + (c & 1) * 2 * kImgBytes;
#pragma unroll
      for (int ks = 0; ks < kMC / 16; ++ks) {
        bf16x8 fa[2], fb[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[t] = frag(buf, offA[t], ks);
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = frag(buf, offB[t], ks);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (wk == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], ones, accb[i], 0, 0, 0);
        }
      }
      if (c + 1 < n_chunks) {
        write_stage((c + 1) & 1);  // buffer (c + 1) & 1 was last read in iteration c - 1: every wave passed the barrier since
        __syncthreads();
        if (c + 2 < n_chunks) issue_loads(c + 2);
      }
    }
  }

  // ---- partial tile -> workspace, natural [n][k] order (C layout: column on the lane, rows in the registers)
  float* wsw = a.ws_w + ((size_t)cb * a.splits + split) * (kTile * kTile);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int k = 128 * wk + 32 * j + (lane & 31);
        wsw[n * kTile + k] = acc[i][j][r];
      }
  if (wk == 0 && cb_k == 0 && (lane & 31) == 0) {
    float* wsb = a.ws_b + ((size_t)cb_n * a.splits + split) * kTile;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) wsb[64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = accb[i][r];
  }
}

// ---- fp32 operands (the fp32 train step): the same decomposition on v_mfma_f32_32x32x2_f32 (exact fp32 products, 1/16 of the
// bf16 rate: MFMA-bound here, 2 M N K / 157 TFLOP/s).  An fp32 MFMA operand is ONE value per lane -- A[row = lane & 31][k =
// lane >> 5] -- so the fragments are plain ds_read_b32 of the row-major image (32 consecutive floats per half-wave: conflict-
// free, no swizzle, no transpose).  Chunks of 32 tokens (two 32-KiB images per buffer).
constexpr int kMC32 = 32;
constexpr int kRowBytes32 = kTile * 4;
constexpr int kImgBytes32 = kMC32 * kRowBytes32;  // 32 KiB
static_assert(2 * kImgBytes32 / 16 / kThreadsW == kPieces, "same staging shape as the bf16 kernel");

__global__ __launch_bounds__(kThreadsW) void token_wgrad_f32_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2 buffers][dY image | X image]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, cb = blockIdx.y;
  const int cb_n = cb / a.cbk, cb_k = cb - cb_n * a.cbk;
  const long long m_begin = (long long)split * a.chunks_per_split * kMC32;
  long long m_end = m_begin + (long long)a.chunks_per_split * kMC32;
  if (m_end > a.M) m_end = a.M;
  const int n_chunks = m_begin < m_end ? (int)((m_end - m_begin + kMC32 - 1) / kMC32) : 0;
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)(a.M * a.N * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.M * a.K * 4), 0x00020000);

  // staging: piece id = tid + 512 i (i < 4) of each operand: row = id >> 6, 16-byte piece = id & 63
  int st_row[kPieces / 2], st_pos[kPieces / 2];
  unsigned st_coln[kPieces / 2], st_colk[kPieces / 2];
#pragma unroll
  for (int i = 0; i < kPieces / 2; ++i) {
    const int id = tid + kThreadsW * i;
    st_row[i] = id >> 6;
    st_pos[i] = id & 63;
    const int coln = cb_n * kTile + st_pos[i] * 4, colk = cb_k * kTile + st_pos[i] * 4;
    st_coln[i] = coln < a.N ? (unsigned)(coln * 4) : kOobW;
    st_colk[i] = colk < a.K ? (unsigned)(colk * 4) : kOobW;
  }
  u32x4 stage[kPieces];
  auto issue_loads = [&](int chunk) __attribute__((always_inline)) {
    const long long m0 = m_begin + (long long)chunk * kMC32;
#pragma unroll
    for (int i = 0; i < kPieces / 2; ++i) {
      const long long m = m0 + st_row[i];
      const bool ok = m < m_end;
      const unsigned on = (ok && st_coln[i] != kOobW) ? (unsigned)(m * a.N * 4) + st_coln[i] : kOobW;
      const unsigned ok_ = (ok && st_colk[i] != kOobW) ? (unsigned)(m * a.K * 4) + st_colk[i] : kOobW;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(dy_rs, (int)on, 0, 0);
      stage[kPieces / 2 + i] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)ok_, 0, 0);
    }
  };
  auto write_stage = [&](int buf) __attribute__((always_inline)) {
    unsigned char* base = lds + buf * 2 * kImgBytes32;
#pragma unroll
    for (int i = 0; i < kPieces / 2; ++i) {
      const int off = st_row[i] * kRowBytes32 + st_pos[i] * 16;
      *reinterpret_cast<u32x4*>(base + off) = stage[i];
      *reinterpret_cast<u32x4*>(base + kImgBytes32 + off) = stage[kPieces / 2 + i];
    }
  };

  const int wn = wave & 3, wk = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  int offA[2], offB[4];
#pragma unroll
  for (int t = 0; t < 2; ++t) offA[t] = h * kRowBytes32 + (64 * wn + 32 * t + r) * 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) offB[t] = kImgBytes32 + h * kRowBytes32 + (128 * wk + 32 * t + r) * 4;

  f32x16 acc[2][4], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    accb[i] = f32x16{};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{};
  }

  if (n_chunks > 0) {
    issue_loads(0);
    write_stage(0);
    __syncthreads();
    if (n_chunks > 1) issue_loads(1);
    for (int c = 0; c < n_chunks; ++c) {
      const unsigned char* buf = lds + (c & 1) * 2 * kImgBytes32;
#pragma unroll 4
      for (int ks = 0; ks < kMC32 / 2; ++ks) {
        float fa[2], fb[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[t] = *reinterpret_cast<const float*>(buf + offA[t] + 2 * ks * kRowBytes32);
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = *reinterpret_cast<const float*>(buf + offB[t] + 2 * ks * kRowBytes32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (wk == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], 1.0f, accb[i], 0, 0, 0);
        }
      }
      if (c + 1 < n_chunks) {
        write_stage((c + 1) & 1);
        __syncthreads();
        if (c + 2 < n_chunks) issue_loads(c + 2);
      }
    }
  }

  float* wsw = a.ws_w + ((size_t)cb * a.splits + split) * (kTile * kTile);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = 64 * wn + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        const int k = 128 * wk + 32 * j + (lane & 31);
        wsw[n * kTile + k] = acc[i][j][q];
      }
  if (wk == 0 && cb_k == 0 && (lane & 31) == 0) {
    float* wsb = a.ws_b + ((size_t)cb_n * a.splits + split) * kTile;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) wsb[64 * wn + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)] = accb[i][q];
  }
}

// dW[n][k] = sum over splits of the partial tiles, db likewise -- in a FIXED order (deterministic): a workgroup owns 16
// consecutive float4 of one dW row; its 16 thread rows each add every 16th split in ascending order, and the 16 sums are added
// in thread-row order through LDS.  (One thread per output looping over 256 partials 256 KiB apart read at 0.7 TB/s.)
constexpr int kRedGroups = 16, kRedCols = 16;  // 256 threads = 16 float4 positions x 16 split groups
__global__ __launch_bounds__(256) void token_wgrad_reduce_kernel(const float* __restrict__ ws_w, const float* __restrict__ ws_b,
                                                                  float* __restrict__ dw, float* __restrict__ db, int N, int K,
                                                                  int cbk, int splits) {
  __shared__ f32x4 part[kRedGroups][kRedCols];
  const int tx = threadIdx.x & (kRedCols - 1), ty = threadIdx.x / kRedCols;
  const int kq = K / 4, segs = (kq + kRedCols - 1) / kRedCols;  // 16-float4 segments per row
  const int n = blockIdx.x / segs, k4 = (blockIdx.x - n * segs) * kRedCols + tx;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (n < N && k4 < kq) {
    const int k = k4 * 4;
    const int cb = (n / kTile) * cbk + k / kTile;
    const float* p = ws_w + (size_t)cb * splits * (kTile * kTile) + (n % kTile) * kTile + (k % kTile);
    // four loads in flight per thread (one at a time left the kernel latency-bound at 1.1 TB/s); the sum keeps a fixed order
    int i = ty;
    for (; i + 3 * kRedGroups < splits; i += 4 * kRedGroups) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(p + (size_t)i * (kTile * kTile));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(p + (size_t)(i + kRedGroups) * (kTile * kTile));
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(p + (size_t)(i + 2 * kRedGroups) * (kTile * kTile));
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(p + (size_t)(i + 3 * kRedGroups) * (kTile * kTile));
      s += a0;
      s += a1;
      s += a2;
      s += a3;
    }
    for (; i < splits; i += kRedGroups) s += *reinterpret_cast<const f32x4*>(p + (size_t)i * (kTile * kTile));
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N && k4 < kq) {
    f32x4 t = part[0][tx];
#pragma unroll
    for (int g = 1; g < kRedGroups; ++g) t += part[g][tx];
    *reinterpret_cast<f32x4*>(dw + (size_t)n * K + k4 * 4) = t;
  }
  // bias: the first ceil(N / 16) workgroups, 16 features each; thread row ty adds every 16th split in ascending order (its loads
  // independent of each other), the 16 rows are added in row order through LDS.  (One thread per feature walking all the splits
  // one dependent load at a time made the workgroups that carried the bias the kernel's critical path: 52 us for a 15-us pass.)
  if (db && (long long)blockIdx.x * kRedCols < N) {
    __shared__ float bpart[kRedGroups][kRedCols];
    const int nb = blockIdx.x * kRedCols + tx;
    float sb = 0.f;
    if (nb < N) {
      const float* p = ws_b + (size_t)(nb / kTile) * splits * kTile + (nb % kTile);
      for (int i = ty; i < splits; i += kRedGroups) sb += p[(size_t)i * kTile];
    }
    bpart[ty][tx] = sb;
    __syncthreads();
    if (ty == 0 && nb < N) {
      float t = bpart[0][tx];
#pragma unroll
      for (int g = 1; g < kRedGroups; ++g) t += bpart[g][tx];
      db[nb] = t;
    }
  }
}

struct WgradPlan {
  int cbn, cbk, splits, cps;
};
static WgradPlan plan(int64_t M, int N, int K, int mc = kMC) {
  WgradPlan p;
  p.cbn = (N + kTile - 1) / kTile;
  p.cbk = (K + kTile - 1) / kTile;
  const int64_t chunks = (M + mc - 1) / mc;
  int64_t want = 256 / (p.cbn * p.cbk);  // about one workgroup per CU
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  p.cps = (int)((chunks + want - 1) / want);
  p.splits = (int)((chunks + p.cps - 1) / p.cps);
  return p;
}

}  // namespace
}  // namespace wm2f

using namespace wm2f;

// (the bf16 and fp32 kernels split M into the same number of partial tiles when M >= 256 * 64; the larger of the two plans)
extern "C" int64_t wm2f_token_wgrad_workspace(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int64_t best = 0;
  for (int mc : {kMC, kMC32}) {
    const WgradPlan p = plan(M, N, K, mc);
    const int64_t b = ((int64_t)p.cbn * p.cbk * p.splits * kTile * kTile + (int64_t)p.cbn * p.splits * kTile) * 4;
    best = b > best ? b : best;
  }
  return best;
}

extern "C" int wm2f_token_wgrad_bf16(const void* dy, const void* x, void* dw, void* db, void* workspace, int64_t M, int N,
                                     int K, void* stream) {
  const char* who = "wm2f_token_wgrad_bf16";
  WM2F_REQUIRE(dy && x && dw && workspace, "%s: null pointer", who);
  WM2F_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "%s: M = %lld, N = %d, K = %d (N and K must be multiples of 8)", who,
               (long long)M, N, K);
  WM2F_REQUIRE(M * N * 2 < 0x7fffffffLL && M * K * 2 < 0x7fffffffLL, "%s: operands beyond 2 GiB (32-bit buffer offsets)", who);
  WM2F_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dw & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
               "%s: pointers must be 16-byte aligned", who);
  const WgradPlan p = plan(M, N, K);
  WgradArgs a;
  a.dy = dy;
  a.x = x;
  a.ws_w = (float*)workspace;
  a.ws_b = a.ws_w + (size_t)p.cbn * p.cbk * p.splits * kTile * kTile;
  a.M = M;
  a.N = N;
  a.K = K;
  a.cbn = p.cbn;
  a.cbk = p.cbk;
  a.splits = p.splits;
  a.chunks_per_split = p.cps;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)token_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kImgBytes) != hipSuccess) {
      set_error("%s: cannot reserve %d bytes of LDS", who, 4 * kImgBytes);
      return WM2F_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(token_wgrad_bf16_kernel, dim3(p.splits, p.cbn * p.cbk), dim3(kThreadsW), 4 * kImgBytes, (hipStream_t)stream, a);
  WM2F_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(token_wgrad_reduce_kernel, dim3((unsigned)(N * ((K / 4 + kRedCols - 1) / kRedCols))), dim3(256), 0, (hipStream_t)stream, a.ws_w,
                     a.ws_b, (float*)dw, (float*)db, N, K, p.cbk, p.splits);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

extern "C" int wm2f_token_wgrad_f32(const void* dy, const void* x, void* dw, void* db, void* workspace, int64_t M, int N,
                                    int K, void* stream) {
  const char* who = "wm2f_token_wgrad_f32";
  WM2F_REQUIRE(dy && x && dw && workspace, "%s: null pointer", who);
  WM2F_REQUIRE(M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0, "%s: M = %lld, N = %d, K = %d (N and K must be multiples of 4)", who,
               (long long)M, N, K);
  WM2F_REQUIRE(M * N * 4 < 0x7fffffffLL && M * K * 4 < 0x7fffffffLL, "%s: operands beyond 2 GiB (32-bit buffer offsets)", who);
  WM2F_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dw & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
               "%s: pointers must be 16-byte aligned", who);
  const WgradPlan p = plan(M, N, K, kMC32);
  WgradArgs a;
  a.dy = dy;
  a.x = x;
  a.ws_w = (float*)workspace;
  a.ws_b = a.ws_w + (size_t)p.cbn * p.cbk * p.splits * kTile * kTile;
  a.M = M;
  a.N = N;
  a.K = K;
  a.cbn = p.cbn;
  a.cbk = p.cbk;
  a.splits = p.splits;
  a.chunks_per_split = p.cps;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)token_wgrad_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kImgBytes32) != hipSuccess) {
      set_error("%s: cannot reserve %d bytes of LDS", who, 4 * kImgBytes32);
      return WM2F_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(token_wgrad_f32_kernel, dim3(p.splits, p.cbn * p.cbk), dim3(kThreadsW), 4 * kImgBytes32, (hipStream_t)stream, a);
  WM2F_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(token_wgrad_reduce_kernel, dim3((unsigned)(N * ((K / 4 + kRedCols - 1) / kRedCols))), dim3(256), 0, (hipStream_t)stream, a.ws_w,
                     a.ws_b, (float*)dw, (float*)db, N, K, p.cbk, p.splits);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
