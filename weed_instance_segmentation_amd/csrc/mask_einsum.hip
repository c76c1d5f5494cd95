// K3: per-query mask-embedding x pixel-feature contraction on the fp32 matrix cores.
// Replaces torch.einsum("bqc,bchw->bqhw"), transformers modeling_mask2former.py:2046.
//
//   emb (B, Q, C)   pix (B, C, HW)   out (B, Q, HW)          all fp32, row-major
//
// MFMA shape: v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/SIMD).  Q = 100 pads to 7 row tiles
// (112) instead of the 128 a 32x32 tile would need.
//
// One wave owns a strip of 64 consecutive pixels (4 column tiles) and ALL query row tiles:
//   - B operand (pix) goes HBM -> registers directly, each element read exactly once by exactly
//     one wave: a lane loads float4 pix[c][p0+4*(lane&15) ..+3]; component j feeds column tile j,
//     so a wave-load is 4 channel rows x 256 contiguous bytes and the 4 column tiles of a lane
//     are 4 consecutive pixels -> the epilogue stores float4 rows.
//   - A operand (emb, <= 112 KiB) is staged once per workgroup into LDS in FRAGMENT order
//     [row tile][k super-step][lane][4], so the main loop's ds_read_b128 is lane-linear
//     (conflict-free) and yields the A values of 4 consecutive k-steps.
//   - k order: super-step s covers channels 16s..16s+15; lane group g = lane>>4 takes channels
//     16s+4g+t at k-step t.  Summation order is a fixed permutation of ascending channel order.
//
// Roofline (fp32): MFMA.  2*B*Q*C*HW flop per call (26.84 GFLOP at config 2) against
// 157.3 TFLOP/s; the Q=100 -> 112 padding caps useful MFMA work at 89 %.  HBM side:
// 4*(B*C*HW + B*Q*HW + B*Q*C) bytes, each read / written once.
//
// Two epilogues (EPI):
//   0  logits: float4 rows, as described above (wm2f_mask_einsum_fwd);
//   1  attention mask (wm2f_mask_einsum_attn_mask_fwd): the thresholded bits of HF:2051-2053 straight from the
//      accumulators -- `sigmoid(logit) < 0.5` as one byte per (query, key), 4 bytes per lane and row, plus the
//      "any key open" flag of the row (HF:1912-1914) -- the logits themselves are never written.  Used where the
//      prediction only feeds the next layer's mask: the einsum then runs on the mask features ALREADY resized to
//      that level (resize and einsum commute), so no resize is left for the epilogue.
// Work split: one wave = one 64-pixel strip x MT query row tiles.  At the mask-feature resolution MT covers all
// queries (the pixel operand is read once); at the small level resolutions (1024 ... 16384 pixels per image) the
// queries are split into more chunks of fewer row tiles so that the launch still has about two waves per SIMD: the
// strip is then re-read by the other chunks' waves through L2, which those sizes afford.
#include "common.h"
#include <stdlib.h>

namespace wm2f {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

constexpr int kStripPix = 64;   // pixels per wave strip (4 column tiles x 16)
constexpr int kEinsumWaves = 8; // most waves per workgroup (2 per SIMD); small-chunk launches use 4

// the dependency's threshold, HF:2051-2053: sigmoid in fp32, then `< 0.5` (NOT `logit < 0`: for -6e-8 < logit < 0 the
// fp32 sigmoid rounds to 0.5 and the key stays open)
__device__ __forceinline__ unsigned blocked_bit(float v) { return (1.f / (1.f + expf(-v))) < 0.5f ? 1u : 0u; }

// MT row tiles run on the matrix cores; REM (0 or 4) further query rows run on the VALU pipe, which is
// otherwise idle under the MFMAs: Q = 100 = 6*16 + 4 would waste 75 % of a 7th row tile.
template <int MT, int REM, int EPI>
__global__ __launch_bounds__(kEinsumWaves* kWave) void mask_einsum_fwd_kernel(const float* __restrict__ emb,
                                                                              const float* __restrict__ pix,
                                                                              void* __restrict__ out_any,
                                                                              int* __restrict__ row_open, int Q, int C,
                                                                              int HW, int q_chunks, int k_valid, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float e_lds[];  // [MT][C/16][64][4], then [C/16][4][4][4] (REM)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_thr = blockDim.x, n_waves = n_thr >> 6;
  // dbg bit 6 (set by the launcher, not a probe): the query chunk is the FAST grid index, so the chunks of one strip run
  // together and share its pixel rows through L2 (the backward's 256-row left operand needs 4 chunks per strip)
  const bool chunk_fast = (dbg & 64) != 0;
  const int b = chunk_fast ? blockIdx.y : blockIdx.y / q_chunks;
  const int chunk = chunk_fast ? blockIdx.x % q_chunks : blockIdx.y % q_chunks;
  const int strip_block = chunk_fast ? blockIdx.x / q_chunks : blockIdx.x;
  const int q0 = chunk * (MT * 16 + REM);
  const int S16 = C / 16, C4 = C / 4;
  float* out = reinterpret_cast<float*>(out_any);

  // ---- stage emb rows q0 .. q0+16*MT-1 into LDS, fragment order, zero rows beyond Q
  const float* eb = emb + (int64_t)b * Q * C;
  for (int idx = tid; idx < MT * 16 * C4; idx += n_thr) {
    const int r = idx / C4, c4 = idx - r * C4;
    const int col = c4 * 4, s = col >> 4, g = (col & 15) >> 2;
    const int mt = r >> 4, m = r & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + r < Q) v = *reinterpret_cast<const float4*>(eb + (int64_t)(q0 + r) * C + col);
    *reinterpret_cast<float4*>(e_lds + ((int64_t)(mt * S16 + s) * 64 + (g * 16 + m)) * 4) = v;
  }
  // remainder rows, laid out [super-step][lane group g][k-step t][row]: a lane reads the 4 rows of its
  // channel 16s+4g+t as one float4 (same address for the 16 lanes of a group -> broadcast)
  float* e_rem = e_lds + MT * 16 * C;
  if (REM) {
    for (int idx = tid; idx < C; idx += n_thr) {  // idx = channel
      float4 v;
      const int qr = q0 + MT * 16;
      v.x = (qr + 0 < Q) ? eb[(int64_t)(qr + 0) * C + idx] : 0.f;
      v.y = (qr + 1 < Q) ? eb[(int64_t)(qr + 1) * C + idx] : 0.f;
      v.z = (qr + 2 < Q) ? eb[(int64_t)(qr + 2) * C + idx] : 0.f;
      v.w = (qr + 3 < Q) ? eb[(int64_t)(qr + 3) * C + idx] : 0.f;
      *reinterpret_cast<float4*>(e_rem + idx * 4) = v;  // channel-major == [s][g][t] order
    }
  }
  __syncthreads();

  const int strip = strip_block * n_waves + wave;
  const int n_strips = ceil_div(HW, kStripPix);
  if (strip >= n_strips) return;  // whole wave leaves together; no barrier follows

  const int g = lane >> 4;
  const int p0 = strip * kStripPix + 4 * (lane & 15);
  const bool pvalid = p0 < HW;  // HW % 4 == 0 is checked on the host
  // Buffer descriptors: per-lane byte offset in ONE VGPR (constant over the loop), the row of
  // each k-step as a scalar offset, and the hardware range check drops the strip / row tails
  // (out-of-range loads return 0, out-of-range stores are discarded).
  const uint32_t kOob = 0x80000000u;
  // k_valid <= C rows of the contraction exist in `pix` (k_valid % 4 == 0; C is k_valid rounded up to 16 and emb's row
  // stride, its columns beyond k_valid zero): the backward's d_pix = emb^T x grad contracts over the Q rows of grad.
  const __amdgpu_buffer_rsrc_t pix_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(pix + (int64_t)b * k_valid * HW), 0, k_valid * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc =
      EPI == 0 ? __builtin_amdgcn_make_buffer_rsrc((void*)(out + (int64_t)b * Q * HW), 0, Q * HW * 4, 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<uint8_t*>(out_any) + (int64_t)b * Q * HW), 0,
                                                   Q * HW, 0x00020000);  // mask bytes: one per (query, key)
  const uint32_t voff = pvalid ? (uint32_t)((4 * g * HW + p0) * 4) : kOob;
  // last super-step: lane groups whose 4 rows lie beyond k_valid load zeros (the row is a SCALAR offset, which the
  // range check does not see, so those lanes' vector offset is put out of range instead)
  const uint32_t voff_last = (16 * (S16 - 1) + 4 * g < k_valid) ? voff : kOob;
  const int row_bytes = HW * 4;

  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 rem[4];  // rem[row][pixel j]: partial sums over THIS lane group's channels
#pragma unroll
  for (int r = 0; r < 4; ++r) rem[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Two named register sets (b0 / b1) and a 2x unrolled loop: the loads of super-step s+1 are
  // issued before the MFMAs of super-step s and waited for with a COUNTED vmcnt only when used.
  f32x4 b0[4], b1[4];
  auto load_b = [&](f32x4 (&dst)[4], int s, uint32_t vo) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pix_rsrc, vo, (16 * s + t) * row_bytes, 0));
  };
  auto compute = [&](const f32x4 (&bb)[4], int s) {
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      a[mt] = *reinterpret_cast<const f32x4*>(e_lds + ((mt * S16 + s) * 64 + lane) * 4);
    f32x4 er[4];
    if (REM) {
#pragma unroll
      for (int t = 0; t < 4; ++t) er[t] = *reinterpret_cast<const f32x4*>(e_rem + ((s * 4 + g) * 4 + t) * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][t], bb[t][j], acc[mt][j], 0, 0, 0);
      }
      if (REM && !(dbg & 1)) {  // 16 VALU FMAs per k-step, hidden in the MFMA issue gaps
#pragma unroll
        for (int r = 0; r < 4; ++r) rem[r] = __builtin_elementwise_fma((f32x4){er[t][r], er[t][r], er[t][r], er[t][r]}, bb[t], rem[r]);
      }
    }
  };
  load_b(b0, 0, S16 == 1 ? voff_last : voff);
  int s = 0;
  for (; s + 2 < S16; s += 2) {  // steady state: both register sets are always reloaded
    load_b(b1, s + 1, voff);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the MFMAs it overlaps with
    compute(b0, s);
    load_b(b0, s + 2, s + 3 == S16 ? voff_last : voff);
    __builtin_amdgcn_sched_barrier(0);
    compute(b1, s + 1);
  }
  {  // peeled tail: one or two super-steps left, no further prefetch
    const bool two = (s + 2 == S16);
    if (two) load_b(b1, s + 1, voff_last);
    __builtin_amdgcn_sched_barrier(0);
    compute(b0, s);
    __builtin_amdgcn_sched_barrier(0);
    if (two) compute(b1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: lane holds rows 4g+reg of each row tile, pixels p0..p0+3 (column tiles 0..3).
  // Rows >= Q fall outside out_rsrc's range and are dropped by the hardware (see the note at the store).
  if (EPI == 1) {
    // attention-mask bytes: 4 consecutive keys of one row per lane -> one dword; the row's "any key open" flag is a
    // plain store of 1 by one lane of the row's 16 (every writer writes the same value into the zero-initialised array)
    const uint32_t moff = pvalid ? (uint32_t)p0 : kOob;
    int* ro = row_open + (int64_t)b * Q;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned bits = blocked_bit(acc[mt][0][r]) | (blocked_bit(acc[mt][1][r]) << 8) |
                              (blocked_bit(acc[mt][2][r]) << 16) | (blocked_bit(acc[mt][3][r]) << 24);
        const int row = q0 + mt * 16 + 4 * g + r;
        __builtin_amdgcn_raw_buffer_store_b32(bits, out_rsrc, moff + (uint32_t)(row * HW), 0, 0);
        const unsigned long long open = __builtin_amdgcn_ballot_w64(pvalid && bits != 0x01010101u);
        if ((lane & 15) == 0 && row < Q && ((open >> (lane & 48)) & 0xffffull)) ro[row] = 1;
      }
    }
    if (REM) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = rem[r][j];
          x += __shfl_xor(x, 16, kWave);
          x += __shfl_xor(x, 32, kWave);
          rem[r][j] = x;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned bits = blocked_bit(rem[r][0]) | (blocked_bit(rem[r][1]) << 8) | (blocked_bit(rem[r][2]) << 16) |
                              (blocked_bit(rem[r][3]) << 24);
        const int row = q0 + MT * 16 + r;
        __builtin_amdgcn_raw_buffer_store_b32(bits, out_rsrc, ((pvalid && g == 0) ? (uint32_t)p0 : kOob) + (uint32_t)(row * HW), 0, 0);
        const unsigned long long open = __builtin_amdgcn_ballot_w64(pvalid && g == 0 && bits != 0x01010101u);
        if (lane == 0 && row < Q && open) ro[row] = 1;
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 v = {acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
      // The row goes into the VECTOR offset: the hardware range check covers voffset + inst_offset only -- an SGPR
      // offset is added unchecked -- so with the row in soffset the padded rows (>= Q) were NOT dropped but written
      // behind the image's slab (into the next image's first rows, or behind the tensor).
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc,
                                             voff + (uint32_t)((q0 + mt * 16 + r) * row_bytes), 0, 0);
      if (dbg & 32) asm volatile("s_nop 3" ::: "memory");
    }
  }
  if (REM && (dbg & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (REM && (dbg & 16)) asm volatile("s_nop 7\n s_nop 7" ::: "memory");
  if (REM && !(dbg & 2)) {  // sum the 4 lane groups' channel partials; group 0 stores rows q0+16*MT .. +3
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = rem[r][j];
        x += __shfl_xor(x, 16, kWave);
        x += __shfl_xor(x, 32, kWave);
        rem[r][j] = x;
      }
    const uint32_t voff_r = (pvalid && g == 0) ? (uint32_t)(p0 * 4) : kOob;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rem[r]), out_rsrc,
                                             voff_r + (uint32_t)((q0 + MT * 16 + r) * row_bytes), 0, 0);
  }
}


// =====================================================================================================
// Backward of the einsum (HF:2046 under autograd):
//   d_pix[b][c][p] = sum_q emb[b][q][c] * grad[b][q][p]     -- the forward kernel itself with the roles
//       (left operand = emb^T, C rows x Q columns, zero-padded to a multiple of 16 columns; right operand = grad,
//       whose rows beyond Q are never addressed: k_valid), 4 chunks of 64 channel rows per strip, chunk-fast grid;
//   d_emb[b][q][c] = sum_p grad[b][q][p] * pix[b][c][p]     -- both operands are contiguous along the contraction:
//       split over pixel ranges, one workgroup = one range x 112 queries x 256 channels (4 waves x 64 channels),
//       partial tiles to the workspace, summed in split order by a second kernel (deterministic, no atomics).

__global__ void emb_transpose_pad_kernel(const float* __restrict__ emb, float* __restrict__ out, int Q, int C, int Qp, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // out (B, C, Qp)
  if (i >= total) return;
  const int q = (int)(i % Qp);
  const int64_t bc = i / Qp;
  const int c = (int)(bc % C);
  const int64_t b = bc / C;
  out[i] = q < Q ? emb[(b * Q + q) * C + c] : 0.f;
}

template <int MT>
__global__ __launch_bounds__(256) void mask_einsum_bwd_emb_kernel(const float* __restrict__ go, const float* __restrict__ pix,
                                                                   float* __restrict__ ws, int Q, int C, int HW, int n_split,
                                                                   int px_per_split, int q_chunks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int split = blockIdx.x, b = blockIdx.y / q_chunks, chunk = blockIdx.y % q_chunks;
  const int q0 = chunk * MT * 16;
  const int n_ct = C / 16;
  const int ct0 = (blockIdx.z * 4 + wave) * 4;  // this wave's 4 column tiles (64 channels)
  if (ct0 >= n_ct) return;                       // whole wave; no barrier in this kernel
  const int p_begin = split * px_per_split;
  const int p_end = min(HW, p_begin + px_per_split);
  if (p_begin >= p_end) return;
  const uint32_t kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t go_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(go + (int64_t)b * Q * HW), 0, Q * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t pix_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(pix + (int64_t)b * C * HW), 0, C * HW * 4, 0x00020000);
  // lane (m, g): row m of the tile, pixels p + 4g .. 4g+3 of the 16-pixel step: component t is k-step t's value
  uint32_t a_voff[MT], b_voff[4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = q0 + 16 * mt + m;
    a_voff[mt] = row < Q ? (uint32_t)((row * HW + 4 * g) * 4) : kOob;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (ct0 + j) * 16 + m;
    b_voff[j] = (ct0 + j < n_ct) ? (uint32_t)((c * HW + 4 * g) * 4) : kOob;
  }
  struct Frag {
    f32x4 a[MT];
    f32x4 bq[4];
  };
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int n_it = ceil_div(p_end - p_begin, 16);
  auto load = [&](Frag& f, int it) __attribute__((always_inline)) {
    const int p = p_begin + 16 * it;
    const bool ok = p + 4 * g < p_end;  // HW % 4 == 0: a lane's 4 pixels are all inside or all outside; steps past the range load zeros
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      f.a[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(go_rsrc, ok ? a_voff[mt] : kOob, p * 4, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      f.bq[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pix_rsrc, ok ? b_voff[j] : kOob, p * 4, 0));
  };
  auto compute = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[mt][t], f.bq[j][t], acc[mt][j], 0, 0, 0);
  };
  Frag f0, f1;
  load(f0, 0);
  for (int it = 0; it < n_it; it += 2) {  // a step past the range computes on zeros (at most one per workgroup)
    load(f1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(f0);
    __builtin_amdgcn_sched_barrier(0);
    load(f0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(f1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // acc[mt][j]: lane (n = m, rows 4g+r) = (query q0 + 16 mt + 4g + r, channel 16 (ct0 + j) + n)
  float* w = ws + (((int64_t)blockIdx.y * n_split + split) * (MT * 16)) * C;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ct0 + j >= n_ct) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[(int64_t)(16 * mt + 4 * g + r) * C + (ct0 + j) * 16 + m] = acc[mt][j][r];
    }
}

__global__ void mask_einsum_bwd_emb_reduce_kernel(const float* __restrict__ ws, float* __restrict__ g_emb, int Q, int C, int n_split,
                                                  int q_chunks, int rows_per_chunk, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // g_emb (B, Q, C), one float4 per thread
  if (i >= total) return;
  const int C4 = C / 4;
  const int c4 = (int)(i % C4);
  const int64_t bq = i / C4;
  const int q = (int)(bq % Q);
  const int64_t b = bq / Q;
  const int chunk = q / rows_per_chunk, r = q - chunk * rows_per_chunk;
  const float* w = ws + (((b * q_chunks + chunk) * n_split) * rows_per_chunk + r) * C + 4 * c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < n_split; ++k) {
    const float4 x = *reinterpret_cast<const float4*>(w + (int64_t)k * rows_per_chunk * C);
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  *reinterpret_cast<float4*>(g_emb + (bq * C) + 4 * c4) = s;
}

}  // namespace wm2f

using namespace wm2f;

namespace {

// epi 0: out = logits (B, Q, HW) fp32.  epi 1: out = attention-mask bytes (B, Q, HW), row_open (B, Q) int32.
int launch_einsum(const void* emb, const void* pix, void* out, int* row_open, int B, int Q, int C, int HW, int epi,
                  void* stream, const char* who, int k_valid = 0, int force_mt = 0) {
  if (k_valid <= 0) k_valid = C;
  WM2F_REQUIRE(k_valid <= C && k_valid > C - 16 && k_valid % 4 == 0, "%s: contraction rows %d against C = %d", who, k_valid, C);
  WM2F_REQUIRE(emb && pix && out && (epi == 0 || row_open), "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && Q > 0 && C > 0 && HW > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(C % 16 == 0, "%s: C=%d must be a multiple of 16", who, C);
  WM2F_REQUIRE(HW % 4 == 0, "%s: HW=%d must be a multiple of 4", who, HW);
  WM2F_REQUIRE((int64_t)C * HW * 4 < (1ll << 31) && (int64_t)(Q + 16) * HW * 4 < (1ll << 31),
               "%s: one image's pix / out slab must stay below 2 GiB (32-bit buffer offsets)", who);
  // Query rows per pass: up to 7 MFMA row tiles (accumulator registers / 160 KiB of LDS); when the last
  // chunk leaves 1..4 rows over a multiple of 16 they go to the VALU side path instead of a padded tile.
  int mt_cap = (160 * 1024 - 16 * C) / (16 * C * 4);
  if (mt_cap > 7) mt_cap = 7;
  WM2F_REQUIRE(mt_cap >= 1, "%s: C=%d too large for the LDS-resident emb tile", who, C);
  const int n_strips = ceil_div(HW, kStripPix);
  int q_chunks = ceil_div(Q, 16 * mt_cap);
  int rows_per_chunk = ceil_div(Q, q_chunks);
  int MT = rows_per_chunk / 16, REM = 0;
  const int left = rows_per_chunk - MT * 16;
  // History: with the row offset in the stores' SGPR offset (see the epilogue) padded rows were written out of bounds
  // and, at C = 64, launches of this remainder variant lost a few lanes of one main-tile store; both are gone with
  // the row in the vector offset.
#ifdef WM2F_PROFILING
  const char* e_dbg = getenv("WM2F_K3_DBG");
  const int dbg = e_dbg ? atoi(e_dbg) : 0;  // probe knob (profiling build only): 1 skips the remainder FMAs, 2 the remainder epilogue
#else
  const int dbg = 0;
#endif
  if (left > 0 && left <= 4 && MT >= 1 && MT <= 6 && Q == q_chunks * rows_per_chunk) REM = 4;  // exact split only
  else if (left > 0) MT += 1;
  if (MT > mt_cap) {  // fall back to plain padding with one more chunk
    q_chunks += 1;
    MT = ceil_div(ceil_div(Q, q_chunks), 16);
    REM = 0;
  }
  // Small launches (the level-resolution predictions: 16 ... 256 strips per image): with all queries in one wave the
  // grid is a handful of workgroups, each staging the whole 100-KiB embedding tile.  Split the queries into chunks of
  // 4 / 2 / 1 row tiles until the launch has about two waves per SIMD (2048); the strip's pixel operand is then read
  // once per chunk, through L2 (<= 7 x 34 MB at these sizes).
  int waves_per_wg = kEinsumWaves;
  {
    const int64_t target = 2048;
    const int64_t units = (int64_t)n_strips * B * q_chunks;
    if (units < target) {
      for (int mt : {4, 2, 1}) {
        if (mt >= MT + (REM ? 1 : 0)) continue;
        const int qc = ceil_div(Q, 16 * mt);
        MT = mt;
        REM = 0;
        q_chunks = qc;
        if ((int64_t)n_strips * B * qc >= target) break;
      }
      waves_per_wg = 4;
    }
  }
  int dbg_flags = dbg;
  if (force_mt > 0) {  // the backward's d_pix launch: exact chunks of force_mt row tiles, chunk-fast grid
    MT = force_mt;
    REM = 0;
    q_chunks = ceil_div(Q, 16 * MT);
    waves_per_wg = kEinsumWaves;
    dbg_flags |= 64;
  }
  while (waves_per_wg > 1 && waves_per_wg / 2 >= n_strips) waves_per_wg /= 2;
  dim3 grid(ceil_div(n_strips, waves_per_wg), B * q_chunks);
  if (dbg_flags & 64) grid = dim3(ceil_div(n_strips, waves_per_wg) * q_chunks, B);
  const size_t lds = (size_t)MT * 16 * C * 4 + (REM ? (size_t)C * 16 : 0);
  hipStream_t st = (hipStream_t)stream;
  if (epi == 1) {
    hipError_t e = hipMemsetAsync(row_open, 0, (size_t)B * Q * sizeof(int), st);
    if (e != hipSuccess) {
      set_error("%s: clearing row_open failed: %s", who, hipGetErrorString(e));
      return WM2F_ELAUNCH;
    }
  }
#define WM2F_LAUNCH_E(MTv, REMv, EPIv)                                                                       \
  if (MT == MTv && REM == REMv && epi == EPIv) {                                                             \
    auto kfn = mask_einsum_fwd_kernel<MTv, REMv, EPIv>;                                                      \
    if (lds > 64 * 1024) {                                                                                   \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) {                                                                                 \
        set_error("%s: cannot raise dynamic LDS to %zu: %s", who, lds, hipGetErrorString(e));               \
        return WM2F_ELAUNCH;                                                                                 \
      }                                                                                                      \
    }                                                                                                        \
    hipLaunchKernelGGL(kfn, grid, dim3(waves_per_wg* kWave), lds, st, (const float*)emb, (const float*)pix, out,  \
                       row_open, Q, C, HW, q_chunks, k_valid, dbg_flags);                                                 \
    launched = true;                                                                                         \
  }
#define WM2F_LAUNCH(MTv, REMv) WM2F_LAUNCH_E(MTv, REMv, 0) WM2F_LAUNCH_E(MTv, REMv, 1)
  bool launched = false;
  WM2F_LAUNCH(1, 0) WM2F_LAUNCH(2, 0) WM2F_LAUNCH(3, 0) WM2F_LAUNCH(4, 0) WM2F_LAUNCH(5, 0) WM2F_LAUNCH(6, 0)
  WM2F_LAUNCH(7, 0) WM2F_LAUNCH(1, 4) WM2F_LAUNCH(2, 4) WM2F_LAUNCH(3, 4) WM2F_LAUNCH(4, 4) WM2F_LAUNCH(5, 4)
  WM2F_LAUNCH(6, 4)
#undef WM2F_LAUNCH
#undef WM2F_LAUNCH_E
  if (!launched) {
    set_error("%s: internal: MT=%d REM=%d", who, MT, REM);
    return WM2F_EINVAL;
  }
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}

}  // namespace

extern "C" int wm2f_mask_einsum_fwd(const void* emb, const void* pix, void* out, int B, int Q, int C, int HW,
                                    int dtype, void* stream) {
  const char* who = "wm2f_mask_einsum_fwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  return launch_einsum(emb, pix, out, nullptr, B, Q, C, HW, 0, stream, who);
}

extern "C" int wm2f_mask_einsum_attn_mask_fwd(const void* emb, const void* pix, void* mask, void* row_open, int B, int Q,
                                              int C, int HW, int dtype, void* stream) {
  const char* who = "wm2f_mask_einsum_attn_mask_fwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built", who);
  return launch_einsum(emb, pix, mask, (int*)row_open, B, Q, C, HW, 1, stream, who);
}

namespace {
struct BwdPlan {
  int Qp, mt, q_chunks, n_split, px_per_split, nz;
  int64_t embt_bytes, ws_bytes;
};
BwdPlan bwd_plan(int B, int Q, int C, int HW) {
  BwdPlan p;
  p.Qp = ceil_div(Q, 16) * 16;
  const int q_tiles = ceil_div(Q, 16);
  p.q_chunks = ceil_div(q_tiles, 7);
  p.mt = ceil_div(q_tiles, p.q_chunks);
  p.nz = ceil_div(C / 16, 16);  // 256 channels per workgroup
  int want = ceil_div(512, B * p.q_chunks * p.nz);
  const int max_split = ceil_div(HW, 256);
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  p.px_per_split = ceil_div(ceil_div(HW, want), 16) * 16;
  p.n_split = ceil_div(HW, p.px_per_split);
  p.embt_bytes = (int64_t)B * C * p.Qp * 4;
  p.ws_bytes = (int64_t)B * p.q_chunks * p.n_split * p.mt * 16 * C * 4;
  return p;
}
}  // namespace

extern "C" int64_t wm2f_mask_einsum_bwd_workspace(int B, int Q, int C, int HW) {
  if (B <= 0 || Q <= 0 || C <= 0 || HW <= 0 || C % 16 != 0) return 0;
  const BwdPlan p = bwd_plan(B, Q, C, HW);
  return p.embt_bytes + p.ws_bytes;
}

extern "C" int wm2f_mask_einsum_bwd(const void* emb, const void* pix, const void* grad_out, void* g_emb, void* g_pix,
                                    void* workspace, int B, int Q, int C, int HW, int dtype, void* stream) {
  const char* who = "wm2f_mask_einsum_bwd";
  WM2F_REQUIRE(dtype == WM2F_F32, "%s: only WM2F_F32 is built (bf16 operands: wm2f_mask_einsum_bf16_bwd)", who);
  WM2F_REQUIRE(emb && pix && grad_out && workspace && (g_emb || g_pix), "%s: null pointer", who);
  WM2F_REQUIRE(B > 0 && Q > 0 && C > 0 && HW > 0, "%s: non-positive size", who);
  if (C % 64 != 0 || Q % 4 != 0 || HW % 4 != 0 || (int64_t)(C + 16) * HW * 4 >= (1ll << 31) ||
      (int64_t)(Q + 16) * HW * 4 >= (1ll << 31)) {
    set_error("%s: shape outside the kernels' (C %% 64 == 0, Q %% 4 == 0, HW %% 4 == 0, one image's slab < 2 GiB): C=%d Q=%d HW=%d",
              who, C, Q, HW);
    return WM2F_EUNSUPPORTED;
  }
  const BwdPlan p = bwd_plan(B, Q, C, HW);
  WM2F_REQUIRE((int64_t)B * p.q_chunks <= 65535, "%s: B exceeds the grid limit", who);
  hipStream_t st = (hipStream_t)stream;
  float* embt = (float*)workspace;
  float* ws = (float*)((char*)workspace + p.embt_bytes);
  if (g_pix) {
    const int64_t total = (int64_t)B * C * p.Qp;
    hipLaunchKernelGGL(emb_transpose_pad_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, (const float*)emb, embt, Q, C,
                       p.Qp, total);
    WM2F_CHECK_LAUNCH(who);
    // rows = channels (C, chunks of 64), contraction = queries (Qp columns of embT, Q rows of grad)
    const int rc = launch_einsum(embt, grad_out, g_pix, nullptr, B, C, p.Qp, HW, 0, stream, who, Q, 4);
    if (rc != WM2F_OK) return rc;
  }
  if (g_emb) {
    dim3 grid(p.n_split, B * p.q_chunks, p.nz);
#define WM2F_BE(MTv)                                                                                                       \
  case MTv:                                                                                                                \
    hipLaunchKernelGGL((mask_einsum_bwd_emb_kernel<MTv>), grid, dim3(256), 0, st, (const float*)grad_out, (const float*)pix, ws, Q, C, \
                       HW, p.n_split, p.px_per_split, p.q_chunks);                                                        \
    break;
    switch (p.mt) {
      WM2F_BE(1) WM2F_BE(2) WM2F_BE(3) WM2F_BE(4) WM2F_BE(5) WM2F_BE(6) WM2F_BE(7)
      default:
        set_error("%s: internal: %d row tiles", who, p.mt);
        return WM2F_EINVAL;
    }
#undef WM2F_BE
    WM2F_CHECK_LAUNCH(who);
    const int64_t total = (int64_t)B * Q * (C / 4);
    hipLaunchKernelGGL(mask_einsum_bwd_emb_reduce_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, (const float*)ws,
                       (float*)g_emb, Q, C, p.n_split, p.q_chunks, p.mt * 16, total);
    WM2F_CHECK_LAUNCH(who);
  }
  return WM2F_OK;
}
