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
                                                                              int HW, int q_chunks, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float e_lds[];  // [MT][C/16][64][4], then [C/16][4][4][4] (REM)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_thr = blockDim.x, n_waves = n_thr >> 6;
  const int b = blockIdx.y / q_chunks, chunk = blockIdx.y % q_chunks;
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

  const int strip = blockIdx.x * n_waves + wave;
  const int n_strips = ceil_div(HW, kStripPix);
  if (strip >= n_strips) return;  // whole wave leaves together; no barrier follows

  const int g = lane >> 4;
  const int p0 = strip * kStripPix + 4 * (lane & 15);
  const bool pvalid = p0 < HW;  // HW % 4 == 0 is checked on the host
  // Buffer descriptors: per-lane byte offset in ONE VGPR (constant over the loop), the row of
  // each k-step as a scalar offset, and the hardware range check drops the strip / row tails
  // (out-of-range loads return 0, out-of-range stores are discarded).
  const uint32_t kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t pix_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(pix + (int64_t)b * C * HW), 0, C * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc =
      EPI == 0 ? __builtin_amdgcn_make_buffer_rsrc((void*)(out + (int64_t)b * Q * HW), 0, Q * HW * 4, 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<uint8_t*>(out_any) + (int64_t)b * Q * HW), 0,
                                                   Q * HW, 0x00020000);  // mask bytes: one per (query, key)
  const uint32_t voff = pvalid ? (uint32_t)((4 * g * HW + p0) * 4) : kOob;
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
  auto load_b = [&](f32x4 (&dst)[4], int s) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      dst[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pix_rsrc, voff, (16 * s + t) * row_bytes, 0));
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
  load_b(b0, 0);
  int s = 0;
  for (; s + 2 < S16; s += 2) {  // steady state: both register sets are always reloaded
    load_b(b1, s + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the MFMAs it overlaps with
    compute(b0, s);
    load_b(b0, s + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(b1, s + 1);
  }
  {  // peeled tail: one or two super-steps left, no further prefetch
    const bool two = (s + 2 == S16);
    if (two) load_b(b1, s + 1);
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

}  // namespace wm2f

using namespace wm2f;

namespace {

// epi 0: out = logits (B, Q, HW) fp32.  epi 1: out = attention-mask bytes (B, Q, HW), row_open (B, Q) int32.
int launch_einsum(const void* emb, const void* pix, void* out, int* row_open, int B, int Q, int C, int HW, int epi,
                  void* stream, const char* who) {
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
  while (waves_per_wg > 1 && waves_per_wg / 2 >= n_strips) waves_per_wg /= 2;
  dim3 grid(ceil_div(n_strips, waves_per_wg), B * q_chunks);
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
                       row_open, Q, C, HW, q_chunks, dbg);                                                   \
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
