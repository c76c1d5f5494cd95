// K1, phased quad kernel: multi-scale deformable attention for the encoder's own shape
// (3 levels ordered coarse -> fine with sides 1 : 2 : 4, 4 points, head_dim 32, queries == tokens).
// Same arithmetic as msdeform.hip / msdeform_tiled.hip (transformers modeling_mask2former.py:798-837,
// fused prologue :983-1002); this file changes WHO computes what and WHEN the windows arrive.
//
// What the LDS-window kernel (msdeform_tiled.hip) measured as its two limits, and the answer here:
//  1. Instruction issue in the gather: with 4 lanes per query every lane repeated the whole
//     coordinate / weight arithmetic of all 12 points (84 VALU per lane and point, 16 of them the
//     FMAs).  Here lane j of a query's quad owns point j of every level: it computes that point's
//     window address and four corner weights ONCE, and the quad shares them with DPP quad_perm
//     broadcasts (v_mov_b32_dpp, 5 per point) -- about 37 VALU per lane and point.
//  2. Staging and gather ran back to back (one 149.5-KiB workgroup per CU).  Here every thread
//     keeps the accumulators of its (up to 3) queries in registers and the LEVELS are the phases:
//     all three windows are requested up front by LDS-DMA with a fixed number of requests per
//     wave, and the gather of the coarse level starts behind `s_waitcnt vmcnt(mid + fine)`, the
//     mid level behind `vmcnt(fine)`: two thirds of the gather runs under the staging traffic.
//     Each window is its own __shared__ array so that the compiler's LDS-DMA tracking (alias
//     scopes) orders a window's ds_reads behind that window's requests only.
//  Bank conflicts: a quad reads 64 contiguous bytes per ds_read_b128 and the four quads of a
//  16-lane LDS group would all land on the same 16 banks (pixel stride 128 B); the quads alternate
//  which 64-B half of the pixel they read first, so they spread over all four 16-bank blocks.
//
// Window geometry is compile-time: tile 16 x 16 finest-level pixels, margin 4 -> 14^2, 18^2, 26^2
// pixels x 128 B = 25 + 41 + 85 KiB.
// Points whose 2 x 2 footprint leaves the window take the slow path (global loads), as before.
//
// Roofline: HBM, algorithmic bytes as msdeform.hip (550 502 400 B per launch at config 2).
#include "msdeform_tiled.h"
#include <type_traits>

// Cache-policy bits (gfx940+: 1 = sc0, 2 = nt, 16 = sc1) of the streaming kernel's three global streams: window DMA,
// operand rows, output rows.  Overridable at compile time for A/B builds (tools/kbench.py --lib).
#ifndef WM2F_DMA_AUX
#define WM2F_DMA_AUX 0
#endif
#ifndef WM2F_OP_AUX
#define WM2F_OP_AUX 0
#endif
#ifndef WM2F_ST_AUX
#define WM2F_ST_AUX 0
#endif

namespace wm2f {

namespace {

constexpr int kQF = 16, kQM = 4;
constexpr int kThreads = 512, kWavesQ = kThreads / kWave, kPasses = 3;
[[maybe_unused]] constexpr int kQuads = kThreads / 4;
template <int LV> struct Win {
  static constexpr int side = (kQF >> (2 - LV)) + 2 * kQM + 2;  // 14, 18, 26
  static constexpr int npix = side * side;
  static constexpr int chunks = (npix + 7) / 8;                 // 1-KiB LDS-DMA pieces: 25, 41, 85
  static constexpr int per_wave = (chunks + kWavesQ - 1) / kWavesQ;  // requests per wave: 4, 6, 11
};
static_assert(Win<0>::per_wave == 4 && Win<1>::per_wave == 6 && Win<2>::per_wave == 11, "vmcnt constants below");

struct QuadGeom {
  int W0, H0;  // coarsest level; in an exact pyramid level l is (H0 << l, W0 << l)
  int tiles_x, tiles_y;
  int start[3];
  int a_qstride, b_qstride;
  // streaming kernel, any 3-level pyramid ordered coarse -> fine (sizes that are not multiples of 32 give 25x42 / 50x84 /
  // 100x167 ...): the level sizes, their reciprocals, and 1 / (2 W_fine), 1 / (2 H_fine) for the tile -> query arithmetic
  int W[3], H[3], exact;
  float inv_w[3], inv_h[3], inv_2wf, inv_2hf;
};

// MODE 7 (variant 73): the kernel with in-kernel time stamps (s_memtime, wave 0 of each workgroup),
// read back with wm2f_debug_stamps.  A profiling aid; no other mode touches this buffer.
// The stamp buffer -- the library's only device global -- and every ablation / stamped instantiation exist in the
// PROFILING build alone (libwm2f_prof.so, -DWM2F_PROFILING, include/wm2f_prof.h); the production library carries
// MODE 0 kernels only and no global mutable state.
#ifdef WM2F_PROFILING
constexpr int kStampSlots = 160, kStampGroups = 8192;  // 10 waves x 16 slots per workgroup
__device__ long long g_stamps[kStampGroups * kStampSlots];
#define WM2F_STAMP(k)                                                                  \
  do {                                                                                 \
    if (MODE == 7 && tid == 0 && id < kStampGroups) g_stamps[id * kStampSlots + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define WM2F_STAMP(k) do { } while (0)
#endif

template <int K>
__device__ __forceinline__ float bcast(float v) {  // value of lane K of this lane's quad
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xF, 0xF, true));
}
template <int K>
__device__ __forceinline__ int bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)));
  return fmaxf(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
  return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

constexpr int waitcnt_vm(int n) { return (n & 15) | (0x7 << 4) | (0xF << 8) | ((n >> 4) << 14); }  // vmcnt(n) only

// s_waitcnt vmcnt(N) that survives the compiler's own wait insertion.  While an LDS-DMA is pending,
// SIInsertWaitcnts answers ANY vmcnt need of the next instruction with vmcnt(0) (it treats the DMA
// as a flat access that may complete out of order) and folds that into a directly preceding wait.
// The s_nop needs nothing, so the counted wait is emitted as written and entered in the pass's
// scoreboard; the loads it covers are then known complete when their first user follows.
// Workgroup barrier without the memory-model fence of __syncthreads(): the fence makes the compiler
// drain vmcnt while LDS-DMA is pending.  What the fence would order is ordered here by hand: a
// window is read only behind the counted wait of every wave (wait_vm) plus this barrier, and the
// "memory" clobbers keep the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(waitcnt_vm(N));
  asm volatile("s_nop 0");
  __builtin_amdgcn_sched_barrier(0);
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2q = __attribute__((ext_vector_type(2))) unsigned;
typedef __bf16 bf16x2q_t __attribute__((ext_vector_type(2)));
typedef float f32x2c_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rne(float lo, float hi) {  // v_cvt_pk_bf16_f32
  const f32x2c_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2q_t));
}
using f32x4q = __attribute__((ext_vector_type(4))) float;

struct Acc {
  f32x2 a_lo, a_hi, b_lo, b_hi;  // first-read half (4 channels), second-read half (unused when a lane owns 4 channels)
};

// Request one level's window: a wave-instruction moves 8 pixels x 128 B, lane-linear in LDS.
// EVERY wave issues exactly Win<LV>::per_wave requests, so that the phase waits can count them; the
// few requests beyond the last piece fetch that last piece again (same bytes to the same place).
// (A separate landing pad for them would be cheaper on the memory side, but a conditional LDS-DMA
// into a second LDS object makes the compiler's LDS-DMA tracking give up and drain vmcnt to 0
// before the first window read -- measured on the ISA.)
// Addressing: buffer_load_dwordx4 ... lds through a descriptor of this (image, head, level) slab and a
// 32-bit byte offset per lane; a pixel outside the image gets an offset that fails the descriptor's
// range check, and the hardware then writes zeros to LDS (tools/probes/lds_dma_oob.hip) = zero padding.
constexpr unsigned kOobOffset = 0x80000000u;

template <int LV>
__device__ __forceinline__ void stage_level(float4* win, __amdgpu_buffer_rsrc_t slab, int Wl, int Hl, int wx0, int wy0,
                                            int row_bytes, int wave, unsigned pix_lane, unsigned lane_part) {
  using W = Win<LV>;
#pragma unroll
  for (int i = 0; i < W::per_wave; ++i) {
    int c = wave + kWavesQ * i;
    c = c < W::chunks ? c : W::chunks - 1;
    const unsigned idx = (unsigned)(c * 8) + pix_lane;  // < 1024; the tail of the last piece lands in its padding
    const unsigned wy = idx / (unsigned)W::side, wx = idx - wy * (unsigned)W::side;
    const int x = wx0 + (int)wx, y = wy0 + (int)wy;
    const bool in = ((unsigned)x < (unsigned)Wl) & ((unsigned)y < (unsigned)Hl);
    const unsigned off = in ? __umul24(__umul24((unsigned)y, (unsigned)Wl) + (unsigned)x, (unsigned)row_bytes) + lane_part
                            : kOobOffset;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(slab, (lptr_t)(win + c * 64), 16, (int)off, 0, 0, 0);
  }
}

// What a lane publishes to its quad for one (query, level): its point's top-left corner (byte offset
// in the window) and the four corner weights, attention weight folded in; zeros when the point is
// not inside the window (slow path) or the query slot is empty.
struct Prod {
  int addr;
  float w00, w01, w10, w11;
};

// `base`: 0, or the window's LDS byte address -- then the published address is absolute and a consumer needs one add
// (its lane offset) per half instead of two (streaming kernel; the phased kernel keeps window-relative addresses
// because its window reads must stay visibly based on the window's __shared__ array for the LDS-DMA tracking).
// PB = bytes per window pixel: 128 (the head's 32 channels) or 64 (one 16-channel half).
template <int LV, int PB = 128>
__device__ __forceinline__ Prod produce(float px, float py, float aw, int wx0, int wy0, bool valid, unsigned& slow,
                                        int base = 0) {
  constexpr int WW = Win<LV>::side;
  const float x0f = floorf(px), y0f = floorf(py);
  const int xr = (int)x0f - wx0, yr = (int)y0f - wy0;
  const bool fast = ((unsigned)xr < (unsigned)(WW - 1)) & ((unsigned)yr < (unsigned)(WW - 1));
  const bool use = fast & valid;
  Prod p;
  p.addr = base + (use ? (yr * WW + xr) * PB : 0);
  const float a = use ? aw : 0.f;
  slow |= (valid & !fast) ? (1u << LV) : 0u;
  const float fx1 = px - x0f, fy1 = py - y0f;
  const float a1 = a * fy1, a0 = a - a1;
  p.w01 = a0 * fx1;
  p.w00 = a0 - p.w01;
  p.w11 = a1 * fx1;
  p.w10 = a1 - p.w11;
  return p;
}

// The corner reads of one point: 8 x 16 B (2 halves x 4 corners), or 4 x 16 B when a lane owns 4 channels.  ONE such
// buffer rolls through a phase: as soon as the FMA of corner i has issued, the read of the next point's corner i refills
// the same registers, so 7 to 8 reads stay in flight per wave at a cost of 32 registers (double-buffering whole points
// cost 72 and spilled in the streaming kernel).
struct Corners {
  float4 v[8];
};
typedef const __attribute__((address_space(3))) float4* lds_cf4_t;
struct PointAddr {  // where the quad reads a point, and its four corner weights
  const float4 *c1, *c2;  // window-relative form (win != nullptr)
  int a1, a2;             // absolute LDS byte addresses (win == nullptr: the producer folded the window base in)
  float q[4];
};

template <int LV, int K>
__device__ __forceinline__ PointAddr point_addr(const float4* win, const Prod& p, int off1, int off2) {
  PointAddr a;
  const int ak = bcast<K>(p.addr);
  a.q[0] = bcast<K>(p.w00);
  a.q[1] = bcast<K>(p.w01);
  a.q[2] = bcast<K>(p.w10);
  a.q[3] = bcast<K>(p.w11);
  a.c1 = a.c2 = nullptr;
  a.a1 = a.a2 = 0;
  if (win) {
    const char* base = reinterpret_cast<const char*>(win);
    a.c1 = reinterpret_cast<const float4*>(base + ak + off1);
    a.c2 = reinterpret_cast<const float4*>(base + ak + off2);
  } else {
    a.a1 = ak + off1;
    a.a2 = ak + off2;
  }
  return a;
}

template <int LV, int I, int MODE, bool ABS, int PB = 128>
__device__ __forceinline__ float4 read_corner(const PointAddr& a) {
  constexpr int WW = Win<LV>::side;
  constexpr int o = (I & 1) * (PB / 16) + ((I >> 1) & 1) * WW * (PB / 16);  // corner order: 00, 01, 10, 11 (float4 units)
  if (MODE == 4) {  // ablation: no LDS reads
    if (ABS) asm volatile("" ::"v"(a.a1), "v"(a.a2));
    else asm volatile("" ::"v"(a.c1), "v"(a.c2));
    return make_float4(a.q[0], a.q[1], a.q[2], a.q[3]);
  }
  if (ABS) {  // an LDS address held as an integer; the cast chain keeps the access a ds_read (address-space inference)
    const float4* gp = (const float4*)reinterpret_cast<lds_cf4_t>((size_t)((I < 4 ? a.a1 : a.a2) + o * 16));
    return *gp;
  }
  return (I < 4 ? a.c1 : a.c2)[o];
}

// One level for all kPasses queries of the lane: 12 points (pass t, producer lane k), see Corners.
// NC = corner reads per point and lane: 8 (a lane owns 8 channels: two 16-byte halves x 4 corners) or 4 (4 channels).
template <int LV, int GI, int I, int MODE, bool ABS, int NC, int PB>
__device__ __forceinline__ void corner_steps(Acc& acc, Corners& cr, const PointAddr& cur, const PointAddr& nxt) {
  if constexpr (I < NC) {
    if (I < 4) pk_fma4(acc.a_lo, acc.a_hi, cur.q[I & 3], cr.v[I]);
    else pk_fma4(acc.b_lo, acc.b_hi, cur.q[I & 3], cr.v[I]);
    if constexpr (GI + 1 < kPasses * 4) cr.v[I] = read_corner<LV, I, MODE, ABS, PB>(nxt);
    __builtin_amdgcn_sched_barrier(0);
    corner_steps<LV, GI, I + 1, MODE, ABS, NC, PB>(acc, cr, cur, nxt);
  }
}

template <int LV, int GI, int MODE, bool ABS, int NC, int PB>
__device__ __forceinline__ void pipe_step(const float4* win, Acc (&acc)[kPasses], const Prod (&pr)[kPasses], int off1,
                                          int off2, Corners& cr, const PointAddr& cur, bool skip_last) {
  if constexpr (GI < kPasses * 4) {
    if (GI == (kPasses - 1) * 4 && skip_last) return;  // wave-uniform: the last pass holds no query in this wave
    PointAddr nxt = cur;
    if constexpr (GI + 1 < kPasses * 4) nxt = point_addr<LV, (GI + 1) & 3>(ABS ? nullptr : win, pr[(GI + 1) >> 2], off1, off2);
    __builtin_amdgcn_sched_barrier(0);
    corner_steps<LV, GI, 0, MODE, ABS, NC, PB>(acc[GI >> 2], cr, cur, nxt);
    // Pin the sums here: the accumulators are only stored at the very end, and LLVM's code sinking otherwise
    // moves whole FMA chains down there (every corner then stays live across all three phases: 500+ spills).
    if (NC == 8) asm volatile("" : "+v"(acc[GI >> 2].a_lo), "+v"(acc[GI >> 2].a_hi), "+v"(acc[GI >> 2].b_lo), "+v"(acc[GI >> 2].b_hi));
    else asm volatile("" : "+v"(acc[GI >> 2].a_lo), "+v"(acc[GI >> 2].a_hi));
    pipe_step<LV, GI + 1, MODE, ABS, NC, PB>(win, acc, pr, off1, off2, cr, nxt, skip_last);
  }
}

template <int LV, int I, int MODE, bool ABS, int NC, int PB>
__device__ __forceinline__ void first_reads(Corners& cr, const PointAddr& a) {
  if constexpr (I < NC) {
    cr.v[I] = read_corner<LV, I, MODE, ABS, PB>(a);
    first_reads<LV, I + 1, MODE, ABS, NC, PB>(cr, a);
  }
}

// CH = channels per lane: 8 (a quad covers the head's 32 channels, 128-byte window pixels) or 4 (a quad covers one
// 16-channel half of the head, 64-byte window pixels).
template <int LV, int MODE, bool ABS = false, int CH = 8>
__device__ __forceinline__ void gather_phase(const float4* win, Acc (&acc)[kPasses], const float (&px)[kPasses][3],
                                             const float (&py)[kPasses][3], const float (&wt)[kPasses][3],
                                             const bool (&valid)[kPasses], int wx0, int wy0, unsigned (&slow)[kPasses],
                                             int off1, int off2, bool skip_last) {
  constexpr int NC = CH, PB = CH * 16;
  Prod pr[kPasses];
  const int base = ABS ? (int)(size_t)(lds_cf4_t)win : 0;
#pragma unroll
  for (int t = 0; t < kPasses; ++t)
    pr[t] = produce<LV, PB>(px[t][LV], py[t][LV], wt[t][LV], wx0, wy0, valid[t], slow[t], base);
  Corners cr;
  const PointAddr a0 = point_addr<LV, 0>(ABS ? nullptr : win, pr[0], off1, off2);
  first_reads<LV, 0, MODE, ABS, NC, PB>(cr, a0);
  __builtin_amdgcn_sched_barrier(0);
  pipe_step<LV, 0, MODE, ABS, NC, PB>(win, acc, pr, off1, off2, cr, a0, skip_last);
}

// Slow path for one point and 4 channels: per-corner image-bounds checks, corners from global memory.
__device__ __forceinline__ void quad_point_slow(float4& acc, const float* __restrict__ vlev, int Hl, int Wl,
                                                int row_stride, float x, float y, float aw) {
  if (!(x > -1.f && x < (float)Wl && y > -1.f && y < (float)Hl)) return;
  const float x0f = floorf(x), y0f = floorf(y);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx1 = x - x0f, fy1 = y - y0f, fx0 = 1.f - fx1, fy0 = 1.f - fy1;
  const bool xl = x0 >= 0, xr = x0 + 1 < Wl, yt = y0 >= 0, yb = y0 + 1 < Hl;
  const float* p00 = vlev + (int64_t)(y0 * Wl + x0) * row_stride;
  if (yt && xl) fma4s(acc, aw * fy0 * fx0, ld4g(p00));
  if (yt && xr) fma4s(acc, aw * fy0 * fx1, ld4g(p00 + row_stride));
  if (yb && xl) fma4s(acc, aw * fy1 * fx0, ld4g(p00 + (int64_t)Wl * row_stride));
  if (yb && xr) fma4s(acc, aw * fy1 * fx1, ld4g(p00 + (int64_t)(Wl + 1) * row_stride));
}

#ifdef WM2F_PROFILING  // the phased kernel: superseded by the streaming form below; kept as the measured baseline of it
// FUSED = false: a = loc (B,Q,heads,3,4,2), b = attn_w (B,Q,heads,3,4)
// FUSED = true : a = raw offsets, b = raw logits; reference points are recomputed from the query grid.
// MODE 0 = the kernel; 1 = staging only, 2 = gather only, 4 = gather without LDS reads: timing
// ablations (outputs NOT valid), reachable only through wm2f_msdeform_fwd_v variants 13 / 23 / 43.
template <bool FUSED, int MODE>
__global__ __launch_bounds__(kThreads) void msdeform_quad_fwd_kernel(const float* __restrict__ value,
                                                                     const float* __restrict__ a_in,
                                                                     const float* __restrict__ b_in,
                                                                     float* __restrict__ out, QuadGeom g, int S, int Q,
                                                                     int heads, int n_logical, int per_xcd) {
  constexpr int D = 32, NL = 3, P = 4;
  __shared__ __attribute__((aligned(16))) float4 win0[Win<0>::chunks * 64];
  __shared__ __attribute__((aligned(16))) float4 win1[Win<1>::chunks * 64];
  __shared__ __attribute__((aligned(16))) float4 win2[Win<2>::chunks * 64];
  const int id = xcd_contiguous_id(blockIdx.x, per_xcd);
  if (id >= n_logical) return;
  const int n_tiles = g.tiles_x * g.tiles_y;
  const int h = id % heads, bt = id / heads;  // heads innermost: the 8 heads of a tile share loc / weight lines
  const int tile = bt % n_tiles, b = bt / n_tiles;
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  const int tid = threadIdx.x, j = tid & 3;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Quad -> query and first-read half, chosen for the LDS banks.  A ds_read_b128 is served in 16-lane groups
  // {0-3,12-15,20-27} / {4-11,16-19,28-31}: quads {0,3,5,6} / {1,2,4,7} of each half-wave.  A quad reads 64 B =
  // one of four 16-bank blocks: block = 2 * (pixel & 1) + half.  Group mates that read the same half get
  // NEIGHBOURING queries of a tile row: for an offset field that varies slowly they sample the same pixel
  // (same address: broadcast) or adjacent pixels (other parity), the other pair reads the other half first,
  // so the four quads of a group fall in four different blocks.  (Independent random offsets still collide:
  // 1.75 LDS cycles per group on average, measured 1.71.)
  const int quad = lane >> 2;
  const int xq = (0x73261540 >> ((quad & 7) * 4)) & 7;      // quads 0..7 -> row positions 0,4,5,1,6,2,3,7
  const int slot = (tid >> 6) * 16 + (quad & 8) + xq;
  const int hq = (quad >> 2) & 1;                            // which 64-B half of a pixel this quad reads first
  const int off1 = j * 16 + hq * 64, off2 = j * 16 + (1 - hq) * 64;
  const int row_stride = heads * D;
  const float* vb = value + ((int64_t)b * S * heads + h) * D;  // head slice of token 0
  WM2F_STAMP(0);

  // ---- per-level geometry of this tile (wave-uniform).  Level l has sides (H0 << l, W0 << l).
  int wx0[NL], wy0[NL], qx0[NL], qy0[NL], nqx[NL], qcnt[NL + 1];
  qcnt[0] = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int Wl = g.W0 << l, Hl = g.H0 << l, fq = kQF >> (2 - l);
    qx0[l] = tx * fq;
    qy0[l] = ty * fq;
    wx0[l] = qx0[l] - 1 - kQM;  // floor(first pixel coordinate - 0.5) - margin
    wy0[l] = qy0[l] - 1 - kQM;
    int nx = Wl - qx0[l], ny = Hl - qy0[l];
    nx = nx < 0 ? 0 : (nx > fq ? fq : nx);
    ny = ny < 0 ? 0 : (ny > fq ? fq : ny);
    nqx[l] = nx;
    qcnt[l + 1] = qcnt[l] + nx * ny;
  }
  const int nq = qcnt[NL];

  // ---- this lane's operands: point j of every level, for up to kPasses queries (32-bit buffer offsets)
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_in, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)b_in, 0, 0x7fffffff, 0x00020000);
  float2 lc[kPasses][NL];
  float wt[kPasses][NL];
  float refx[kPasses], refy[kPasses];
  int qrow[kPasses];  // b * Q + q
  bool valid[kPasses];
  const float inv_w0 = __builtin_amdgcn_rcpf((float)g.W0), inv_h0 = __builtin_amdgcn_rcpf((float)g.H0);
#pragma unroll
  for (int t = 0; t < kPasses; ++t) {
    int qi = slot + kQuads * t;
    valid[t] = qi < nq;
    if (!valid[t]) qi = 0;
    const bool ge1 = qi >= qcnt[1], ge2 = qi >= qcnt[2];
    const int lq = (ge1 ? 1 : 0) + (ge2 ? 1 : 0);
    const int nx = ge2 ? nqx[2] : (ge1 ? nqx[1] : nqx[0]);
    const int ox = ge2 ? qx0[2] : (ge1 ? qx0[1] : qx0[0]);
    const int oy = ge2 ? qy0[2] : (ge1 ? qy0[1] : qy0[0]);
    const int loc_i = qi - (ge2 ? qcnt[2] : (ge1 ? qcnt[1] : 0));
    const int nxs = nx < 1 ? 1 : nx;
    const int ly_ = (int)(((float)loc_i + 0.5f) * __builtin_amdgcn_rcpf((float)nxs));  // exact: small integers
    const int lx_ = loc_i - ly_ * nxs;
    const int qxi = ox + lx_, qyi = oy + ly_;
    int q = (ge2 ? g.start[2] : (ge1 ? g.start[1] : 0)) + (int)__umul24((unsigned)qyi, (unsigned)(g.W0 << lq)) + qxi;
    if (q > Q - 1) q = Q - 1;
    qrow[t] = b * Q + q;
    // (q + 0.5) / (W0 * 2^lq): the power of two is exact, so this equals (q + 0.5) * rcp(W_q)
    const float sc = ge2 ? 0.25f : (ge1 ? 0.5f : 1.f);
    refx[t] = ((float)qxi + 0.5f) * (inv_w0 * sc);
    refy[t] = ((float)qyi + 0.5f) * (inv_h0 * sc);
    const int a_off = (qrow[t] * g.a_qstride + h * (NL * P * 2) + j * 2) * 4;
    const int b_off = (qrow[t] * g.b_qstride + h * (NL * P) + j) * 4;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      lc[t][l] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(a_rs, a_off + l * (P * 2 * 4), 0, 0));
      wt[t][l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rs, b_off + l * (P * 4), 0, 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the vmcnt constants below rely on this order: operands, coarse, mid, fine
  WM2F_STAMP(1);

  // ---- 1. request all three windows (4 + 6 + 11 LDS-DMA instructions per wave)
  if (MODE != 2) {
    const unsigned pix_lane = (unsigned)lane >> 3, lane_part = ((unsigned)lane & 7u) * 16u;
    const int row_bytes = row_stride * 4, px0 = g.W0 * g.H0;
    // one descriptor per level: this (image, head)'s slab from the level's first token to its last pixel
    const __amdgpu_buffer_rsrc_t s0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vb + (int64_t)g.start[0] * row_stride), 0, (px0 - 1) * row_bytes + D * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t s1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vb + (int64_t)g.start[1] * row_stride), 0, (4 * px0 - 1) * row_bytes + D * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t s2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vb + (int64_t)g.start[2] * row_stride), 0, (16 * px0 - 1) * row_bytes + D * 4, 0x00020000);
    stage_level<0>(win0, s0, g.W0, g.H0, wx0[0], wy0[0], row_bytes, wave, pix_lane, lane_part);
    __builtin_amdgcn_sched_barrier(0);
    stage_level<1>(win1, s1, g.W0 << 1, g.H0 << 1, wx0[1], wy0[1], row_bytes, wave, pix_lane, lane_part);
    __builtin_amdgcn_sched_barrier(0);
    stage_level<2>(win2, s2, g.W0 << 2, g.H0 << 2, wx0[2], wy0[2], row_bytes, wave, pix_lane, lane_part);
    __builtin_amdgcn_sched_barrier(0);
  }
  WM2F_STAMP(2);
  if (MODE == 1) {  // ablation: staging only
    wait_vm<0>();
    wg_barrier();
    if (tid == 0) out[(int64_t)id] = win0[id & 1023].x + win1[id & 1023].y + win2[id & 1023].z + lc[0][0].x + wt[2][2];
    return;
  }

  // ---- 2. operands + coarse window of THIS wave have landed; mid and fine stay in flight
  if (MODE != 2) wait_vm<Win<1>::per_wave + Win<2>::per_wave>();
  WM2F_STAMP(3);
  // per-point pixel coordinates and (fused) softmax weights, while the other waves' requests land
  float px[kPasses][NL], py[kPasses][NL], sm_max[kPasses], sm_inv[kPasses];
#pragma unroll
  for (int t = 0; t < kPasses; ++t) {
    sm_max[t] = 0.f;
    sm_inv[t] = 1.f;
    if (FUSED) {  // softmax over the 12 logits of the quad (HF:986-991)
      const float mx = quad_max(fmaxf(fmaxf(wt[t][0], wt[t][1]), wt[t][2]));
      float s = 0.f;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        wt[t][l] = __expf(wt[t][l] - mx);
        s += wt[t][l];
      }
      const float inv = __builtin_amdgcn_rcpf(quad_sum(s));
#pragma unroll
      for (int l = 0; l < NL; ++l) wt[t][l] *= inv;
      sm_max[t] = mx;
      sm_inv[t] = inv;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const float Wl = (float)(g.W0 << l), Hl = (float)(g.H0 << l);
      if (FUSED) {  // loc = ref + off / (W, H); pixel = loc * (W, H) - 0.5  ==  ref * W - 0.5 + off
        px[t][l] = (refx[t] * Wl - 0.5f) + lc[t][l].x;
        py[t][l] = (refy[t] * Hl - 0.5f) + lc[t][l].y;
      } else {  // grid_sample's own arithmetic (align_corners = False)
        px[t][l] = ((2.f * lc[t][l].x - 1.f + 1.f) * Wl - 1.f) * 0.5f;
        py[t][l] = ((2.f * lc[t][l].y - 1.f + 1.f) * Hl - 1.f) * 0.5f;
      }
    }
  }
  Acc acc[kPasses];
  unsigned slow[kPasses];
#pragma unroll
  for (int t = 0; t < kPasses; ++t) {
    acc[t].a_lo = acc[t].a_hi = acc[t].b_lo = acc[t].b_hi = (f32x2){0.f, 0.f};
    slow[t] = 0;
  }

  // wave-uniform: no lane of this wave holds a query in the last pass (waves 5-7 of an interior tile)
  const bool skip_last = __builtin_amdgcn_ballot_w64(valid[kPasses - 1]) == 0;
  __builtin_amdgcn_sched_barrier(0);
  WM2F_STAMP(4);
  wg_barrier();  // every wave's coarse requests have landed
  WM2F_STAMP(5);
  gather_phase<0, MODE>(win0, acc, px, py, wt, valid, wx0[0], wy0[0], slow, off1, off2, skip_last);
  WM2F_STAMP(6);

  if (MODE != 2) wait_vm<Win<2>::per_wave>();
  WM2F_STAMP(7);
  wg_barrier();
  WM2F_STAMP(8);
  gather_phase<1, MODE>(win1, acc, px, py, wt, valid, wx0[1], wy0[1], slow, off1, off2, skip_last);
  WM2F_STAMP(9);

  if (MODE != 2) wait_vm<0>();
  WM2F_STAMP(10);
  wg_barrier();
  WM2F_STAMP(11);
  gather_phase<2, MODE>(win2, acc, px, py, wt, valid, wx0[2], wy0[2], slow, off1, off2, skip_last);
  WM2F_STAMP(12);

  // ---- 3. slow points (rare), then the stores
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
  const bool any_slow = (slow[0] | slow[1] | slow[2]) != 0;
  const bool wave_slow = __builtin_amdgcn_ballot_w64(any_slow) != 0;
#pragma unroll
  for (int t = 0; t < kPasses; ++t) {
    if (!valid[t]) continue;
    float4 r1 = make_float4(acc[t].a_lo.x, acc[t].a_lo.y, acc[t].a_hi.x, acc[t].a_hi.y);
    float4 r2 = make_float4(acc[t].b_lo.x, acc[t].b_lo.y, acc[t].b_hi.x, acc[t].b_hi.y);
    if (wave_slow) {
      const unsigned bits = (unsigned)bcast<0>((int)slow[t]) | ((unsigned)bcast<1>((int)slow[t]) << 3) |
                            ((unsigned)bcast<2>((int)slow[t]) << 6) | ((unsigned)bcast<3>((int)slow[t]) << 9);
      unsigned todo = bits;  // bit (k * 3 + l): point k of level l
      const float* ap = a_in + (int64_t)qrow[t] * g.a_qstride + h * (NL * P * 2);
      const float* bp = b_in + (int64_t)qrow[t] * g.b_qstride + h * (NL * P);
      while (todo) {
        const int i = __ffs(todo) - 1;
        todo &= todo - 1;
        const int k = i / 3, l = i - k * 3;
        const int Wl = g.W0 << l, Hl = g.H0 << l;
        const int st_l = l == 0 ? g.start[0] : (l == 1 ? g.start[1] : g.start[2]);
        const float lx = ap[(l * P + k) * 2], ly = ap[(l * P + k) * 2 + 1];
        float aw = bp[l * P + k], x, y;
        if (FUSED) {
          aw = __expf(aw - sm_max[t]) * sm_inv[t];
          x = (refx[t] * (float)Wl - 0.5f) + lx;
          y = (refy[t] * (float)Hl - 0.5f) + ly;
        } else {
          x = ((2.f * lx - 1.f + 1.f) * (float)Wl - 1.f) * 0.5f;
          y = ((2.f * ly - 1.f + 1.f) * (float)Hl - 1.f) * 0.5f;
        }
        const float* vlev = vb + (int64_t)st_l * row_stride;
        quad_point_slow(r1, vlev + (off1 >> 2), Hl, Wl, row_stride, x, y, aw);
        quad_point_slow(r2, vlev + (off2 >> 2), Hl, Wl, row_stride, x, y, aw);
      }
    }
    const int o_off = (qrow[t] * heads + h) * (D * 4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r1), out_rs, o_off + off1, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r2), out_rs, o_off + off2, 0, 0);
  }
  WM2F_STAMP(13);
}


#endif  // WM2F_PROFILING (phased kernel)

// =====================================================================================================
// Streaming form of the kernel above: persistent workgroups, two loader waves.
//
// In-kernel stamps of msdeform_quad_fwd_kernel (profiles/r01_k1_quad_stamps.json): of a workgroup's 31.7k cycles
// the three gather phases take 13.7k; 7.7k go into ISSUING the 168 LDS-DMA requests (the requests are accepted
// at the rate the memory side delivers, ~46 cycles each, and all 8 waves sit in that queue together), 3.1k into
// the per-tile set-up and 4.7k into barriers behind them.  None of that needs the gather waves:
//   * one workgroup per CU stays resident and walks over its tiles (grid = CUs, XCD-contiguous tile ranges);
//   * waves 8 and 9 only move data: they keep the window of the NEXT phase in flight while waves 0-7 gather
//     the current one.  The three windows form the ring: while the coarse window of tile n is gathered the
//     fine window of tile n is requested, under the mid gather the coarse window of tile n + 1, under the fine
//     gather the mid window of tile n + 1.  Hand-over is one s_barrier per phase (10 waves), the loaders join
//     it behind a counted vmcnt that covers exactly the window about to be read;
//   * a loader's per-lane request offsets depend on (level, piece, lane) only: computed once into registers,
//     a request then costs one v_add (tile origin) -- plus a column test on tiles at the left / right image
//     border; rows above / below the image fail the descriptor's range check by themselves;
//   * the gather waves fetch the NEXT tile's locations / weights under the fine gather of the current one,
//     and their decode of "which query is mine" is cached while the tile shape (full / ragged) stays the same.
// The gather waves never see an LDS-DMA in their instruction stream, so the compiler's wait insertion needs
// no help there; the loaders never read LDS.
#ifndef WM2F_STREAM_GATHER_WAVES
#define WM2F_STREAM_GATHER_WAVES 8
#endif
constexpr int kGatherWaves = WM2F_STREAM_GATHER_WAVES;  // 8 or 10 (10: 3 waves on every SIMD at the same 168 registers; measured 3 % slower)
// Two forms of the streaming kernel, by the channels a gather lane owns (CH):
//   CH = 8  a workgroup = (image, head, tile): 128-byte window pixels (149.5 KiB), 8 gather + 2 loader waves, ONE workgroup
//           per CU.  Per-wave stamps (profiles/r02_k1_stream_stamps_*): the three gathers saturate the LDS for about half
//           of a tile's 21k cycles; the other half (operand fetch through the texture-address unit, softmax /
//           coordinates, stores, barrier skew behind the SIMD that carries two 3-pass waves) runs with the LDS idle --
//           and nothing else is resident on the CU to use it.
//   CH = 4  a workgroup = (image, head, 16-channel HALF, tile): 64-byte window pixels (77 KiB), 7 gather waves (7 x 16
//           quads x 3 passes = the tile's 336 queries exactly: no wave idles in the third pass) + 1 loader wave, TWO
//           workgroups per CU (128 registers per wave) that the hardware interleaves freely: one's gathers run under the
//           other's fetch / coordinates / stores.  Same tiles, same windows, same halo; the price is that the two halves
//           each fetch the operands and compute the coordinates of their queries.
//           MEASURED (profiles/r02_pmc_k1_half_vs_full.txt, r02_kbench_k1_half_head.jsonl): 18 % SLOWER than CH = 8
//           (214-220 us against 184 us on the same box).  Every window request now asks L2 for 64-byte half lines:
//           TCP->TCC read requests 16.5 M against 8.2 M per launch, texture-address unit busy 72 % of the kernel against
//           55 %, L2 misses + 24 %, VALU instructions + 21 % -- the request path, not the LDS, is what a second workgroup
//           per CU runs into.  Kept selectable (wm2f_msdeform_fwd_v variant 8) as the measured negative; variant 4 / the
//           production entry points run CH = 8.
template <int CH> struct SCfg {
  static_assert(CH == 8 || CH == 4, "channels per lane");
  static constexpr int PB = CH * 16;         // bytes per window pixel
  static constexpr int PPP = 1024 / PB;      // pixels per 1-KiB LDS-DMA piece
  static constexpr int GW = CH == 8 ? kGatherWaves : 7, NLD = CH == 8 ? 2 : 1;
  static constexpr int THREADS = (GW + NLD) * 64, SPLIT = 8 / CH;  // SPLIT = workgroups per (image, head, tile)
};
template <int LV, int CH> struct SWin {
  static constexpr int chunks = (Win<LV>::npix + SCfg<CH>::PPP - 1) / SCfg<CH>::PPP;    // 25, 41, 85  /  13, 21, 43
  static constexpr int n = (chunks + SCfg<CH>::NLD - 1) / SCfg<CH>::NLD;                // requests per loader
};
template <int LV> struct LWin {
  static constexpr int n = SWin<LV, 8>::n;  // requests per loader: 13, 21, 43 -- in BOTH forms (half the bytes, half the loaders)
};
static_assert(LWin<0>::n == 13 && LWin<1>::n == 21 && LWin<2>::n == 43, "vmcnt constants of the loader");
static_assert(SWin<0, 4>::n == 13 && SWin<1, 4>::n == 21 && SWin<2, 4>::n == 43, "vmcnt constants of the loader, half-head form");

// MODE 7 stamps of the streaming kernel: the workgroup's SECOND tile (steady state); slots 0-9 by wave 0 (gather),
// 10-15 by wave 8 (loader).
#ifdef WM2F_PROFILING
#define WM2F_SSTAMP(slot, who)                                                                              \
  do { /* every wave of the role stamps its own row: [workgroup][wave][slot], second tile of the workgroup */ \
    if (MODE == 7 && k == 1 && lane == 0 && blockIdx.x < kStampGroups)                                      \
      g_stamps[blockIdx.x * kStampSlots + wave * 16 + (slot)] = (long long)__builtin_readcyclecounter();    \
  } while (0)
#else
#define WM2F_SSTAMP(slot, who) do { } while (0)
#endif

struct StreamGeom {
  QuadGeom q;
  int n_logical, per_xcd, wg_per_xcd;
  float inv_heads, inv_ntiles, inv_tiles_x;
  // a workgroup's next tile is `wg_per_xcd` ids further: (step_t tiles, step_h heads) with heads innermost -- or, in SLAB
  // order (tiles innermost: id = (image * heads + head) * n_tiles + tile), step_h slabs and step_t = (step_ty, step_tx) tiles
  int step_t, step_h, step_tx, step_ty;
  // work order of the tiles of one image: vertical strips `strip_w` tiles wide, walked row by row (0 = plain raster).
  // An XCD holds 32 workgroups = 4 tiles x 8 heads at a time and its L2 (4 MiB) about three tiles' windows, so in
  // raster order the 10 halo rows a tile row shares with the next one are long gone when that row comes round
  // (2 x 8 x 1.2 MB later); in 2-wide strips the row below follows 2 tiles later and finds them in L2, and only the
  // strip seams (3 per image instead of 7 tile-row seams) are fetched twice.  Measured (profiles/r02_kbench_k1_tile_order):
  // no faster than raster (165 vs 169 us), so raster stays the default.
  int strip_w, full_strips, rem_w;
  float inv_per_strip, inv_strip_w, inv_rem_w;
  // layout of `value` in floats: pixel pitch, and the head / image strides -- (B, S, heads, 32) or head-major (heads, B, S, 32)
  int v_pix;
  long long v_head, v_img;
};

// exact floor(a / d) for 0 <= a < 2^22 with inv ~ 1/d (one correction step either way)
__device__ __forceinline__ int div_small(int a, int d, float inv) {
  int q = (int)(((float)a + 0.5f) * inv);
  int r = a - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; }
  return q;
}

struct TileId {
  int b, h, tx, ty, hh;  // hh: which 16-channel half of the head (half-head form), else 0
};

// One tile (16 x 16 finest-level pixels) seen from level l: the queries whose reference point (q + 0.5) / n falls into
// the tile -- columns [qx0, qx0 + nqx), rows [qy0, qy0 + nqy) -- and the window origin, floor(first sampled pixel) - margin,
// where a reference point x in [tx 16 / W_f, (tx + 1) 16 / W_f) lands on pixel x W_l - 0.5 of level l.  In an exact
// 1 : 2 : 4 pyramid these are tx * f, f, tx * f - 1 - margin with f = 4 << l; in general they need a division per bound
// (exact float divisions: the host checks the operands stay below 2^22).
struct LevelTile {
  int wx0, wy0, qx0, qy0, nqx, nqy;
};
__device__ __forceinline__ void axis_tile(int t, int n, int nf, float inv_2nf, int& w0, int& q0, int& nq) {
  // first query at or beyond the tile's lower edge: q >= (2 t 16 n - nf) / (2 nf), rounded up (msdeform_tiled.h: q_lo)
  auto lo = [&](int tt) {
    const int a = 2 * tt * kQF * n - nf;
    if (a <= 0) return 0;
    const int v = div_small(a + 2 * nf - 1, 2 * nf, inv_2nf);
    return v > n ? n : v;
  };
  q0 = lo(t);
  nq = lo(t + 1) - q0;
  const int a = 2 * t * kQF * n - nf;  // floor(t 16 n / nf - 0.5) = floor(a / (2 nf)); a < 0 only for t = 0: -1
  w0 = (a < 0 ? -1 : div_small(a, 2 * nf, inv_2nf)) - kQM;
}
__device__ __forceinline__ LevelTile level_tile(const QuadGeom& g, int l, int tx, int ty) {
  LevelTile r;
  if (g.exact || l == 2) {
    const int fq = kQF >> (2 - l);
    r.qx0 = tx * fq;
    r.qy0 = ty * fq;
    r.wx0 = r.qx0 - 1 - kQM;
    r.wy0 = r.qy0 - 1 - kQM;
    const int nx = g.W[l] - r.qx0, ny = g.H[l] - r.qy0;
    r.nqx = nx < 0 ? 0 : (nx > fq ? fq : nx);
    r.nqy = ny < 0 ? 0 : (ny > fq ? fq : ny);
    return r;
  }
  axis_tile(tx, g.W[l], g.W[2], g.inv_2wf, r.wx0, r.qx0, r.nqx);
  axis_tile(ty, g.H[l], g.H[2], g.inv_2hf, r.wy0, r.qy0, r.nqy);
  return r;
}

// Position of a persistent workgroup in its tile sequence (all wave-uniform): image, head, and the tile's index within
// the image in WORK order.  One division chain at the start, then additions.
struct TileWalk {
  int b, h, tile, tx, ty;  // tx, ty: kept incrementally in raster order (no division on the per-tile path)
};
// `split` = workgroups per (image, head, tile): the walk's innermost index is head * split + half
template <int ORDER>  // 0 raster, 1 strips, 2 Z-order
__device__ __forceinline__ void walk_xy(TileWalk& w, const StreamGeom& sg) {  // (tx, ty) of w.tile by division: start-up and strip order
  if (ORDER == 2) {  // Z-order (square power-of-two tile grids): de-interleave the bits of the tile index
    unsigned x = (unsigned)w.tile & 0x5555u, y = ((unsigned)w.tile >> 1) & 0x5555u;
    x = (x | (x >> 1)) & 0x3333u; y = (y | (y >> 1)) & 0x3333u;
    x = (x | (x >> 2)) & 0x0f0fu; y = (y | (y >> 2)) & 0x0f0fu;
    x = (x | (x >> 4)) & 0x00ffu; y = (y | (y >> 4)) & 0x00ffu;
    w.tx = (int)x;
    w.ty = (int)y;
    return;
  }
  if (ORDER != 1 || sg.strip_w <= 0) {
    w.ty = div_small(w.tile, sg.q.tiles_x, sg.inv_tiles_x);
    w.tx = w.tile - w.ty * sg.q.tiles_x;
    return;
  }
  const int per_strip = sg.strip_w * sg.q.tiles_y;
  const int s = div_small(w.tile, per_strip, sg.inv_per_strip);
  if (s >= sg.full_strips) {  // the narrower last strip (tiles_x not a multiple of strip_w)
    const int r = w.tile - sg.full_strips * per_strip;
    w.ty = div_small(r, sg.rem_w, sg.inv_rem_w);
    w.tx = sg.full_strips * sg.strip_w + (r - w.ty * sg.rem_w);
    return;
  }
  const int r = w.tile - s * per_strip;
  w.ty = div_small(r, sg.strip_w, sg.inv_strip_w);
  w.tx = s * sg.strip_w + (r - w.ty * sg.strip_w);
}
template <int ORDER>
__device__ __forceinline__ TileWalk walk_init(int id, const StreamGeom& sg, int heads) {
  TileWalk w;
  const int n_tiles = sg.q.tiles_x * sg.q.tiles_y;
  if (ORDER == 3) {  // slab order: the tiles of one (image, head) slab are consecutive ids
    const int bh = div_small(id, n_tiles, sg.inv_ntiles);
    w.tile = id - bh * n_tiles;
    w.b = div_small(bh, heads, sg.inv_heads);
    w.h = bh - w.b * heads;
    walk_xy<0>(w, sg);
    return w;
  }
  const int bt = div_small(id, heads, sg.inv_heads);
  w.h = id - bt * heads;
  w.b = div_small(bt, n_tiles, sg.inv_ntiles);
  w.tile = bt - w.b * n_tiles;
  walk_xy<ORDER>(w, sg);
  return w;
}
// Next tile of this workgroup: `step_h` heads and `step_t` tiles further.  In raster order the tile coordinates advance with
// a few scalar adds and compares per step (per-wave stamps: the division chains, the kernel-argument reloads they dragged in
// and the waits behind them had made this block 3 - 6 k cycles of a 21 k-cycle tile).
template <int ORDER>
__device__ __forceinline__ void walk_step(TileWalk& w, const StreamGeom& sg, int heads) {
  const int n_tiles = sg.q.tiles_x * sg.q.tiles_y;
  if (ORDER == 3) {  // tile = ty * tiles_x + tx stays true throughout: a wrap of the tile index is a wrap of ty
    w.tile += sg.step_t;
    w.tx += sg.step_tx;
    w.ty += sg.step_ty;
    w.h += sg.step_h;
    if (w.tx >= sg.q.tiles_x) {
      w.tx -= sg.q.tiles_x;
      ++w.ty;
    }
    if (w.tile >= n_tiles) {
      w.tile -= n_tiles;
      w.ty -= sg.q.tiles_y;
      ++w.h;
    }
    while (w.h >= heads) {
      w.h -= heads;
      ++w.b;
    }
    return;
  }
  int adv = sg.step_t;
  w.h += sg.step_h;
  if (w.h >= heads) {
    w.h -= heads;
    ++adv;
  }
  w.tile += adv;
  while (w.tile >= n_tiles) {
    w.tile -= n_tiles;
    ++w.b;
  }
  if (ORDER == 2 || (ORDER == 1 && sg.strip_w > 0)) {
    walk_xy<ORDER>(w, sg);
    return;
  }
  for (int i = 0; i < adv; ++i) {
    if (++w.tx == sg.q.tiles_x) {
      w.tx = 0;
      if (++w.ty == sg.q.tiles_y) w.ty = 0;
    }
  }
}
__device__ __forceinline__ TileId walk_tile(const TileWalk& w, const StreamGeom& sg, int split = 1) {
  TileId t;
  t.b = w.b;
  t.h = split == 2 ? w.h >> 1 : w.h;
  t.hh = split == 2 ? w.h & 1 : 0;
  t.tx = w.tx;
  t.ty = w.ty;
  return t;
}

template <int LV>
struct LoaderRegs {  // per-lane constants of one loader for one level
  unsigned rel[LWin<LV>::n];  // byte offset of this lane's 16 B relative to the window origin pixel
};

template <int LV, int CH>
__device__ __forceinline__ void loader_init(LoaderRegs<LV>& r, int ld, int Wl, int row_bytes, unsigned pix_lane,
                                            unsigned lane_part) {
  using W = Win<LV>;
  constexpr int kLoaders = SCfg<CH>::NLD, kChunks = SWin<LV, CH>::chunks, kPPP = SCfg<CH>::PPP;
#pragma unroll
  for (int i = 0; i < LWin<LV>::n; ++i) {
    int c = ld + kLoaders * i;
    c = c < kChunks ? c : kChunks - 1;
    const unsigned idx = (unsigned)(c * kPPP) + pix_lane;
    const unsigned wy = idx / (unsigned)W::side, wx = idx - wy * (unsigned)W::side;
    r.rel[i] = __umul24(__umul24(wy, (unsigned)Wl) + wx, (unsigned)row_bytes) + lane_part;
  }
}

// Requests [I0, I1) of this loader for one window.  tile_off = byte offset of the window origin pixel in the
// slab (negative above / left of the image); x_border: the window sticks out left or right.
typedef __attribute__((address_space(3))) float4* lds4_t;  // an LDS pointer that never passes through a generic one

// The half-head form keeps only the fine window's table in registers (43 of the 77: its single loader wave lives under the
// 128-register cap of two workgroups per CU) and recomputes the coarse / mid offsets per request (8 VALU each, 34 requests).
template <int LV, int CH> constexpr bool kLoaderTable = (CH == 8) || (LV == 2);

template <int LV, int I0, int I1, int CH>
__device__ __forceinline__ void loader_issue(lds4_t win, const LoaderRegs<LV>& r, __amdgpu_buffer_rsrc_t slab, int ld,
                                             int tile_off, bool x_border, int wx0, int Wl, unsigned pix_lane,
                                             unsigned lane_part = 0, int row_bytes = 0) {
  using W = Win<LV>;
  constexpr int kLoaders = SCfg<CH>::NLD, kChunks = SWin<LV, CH>::chunks, kPPP = SCfg<CH>::PPP;
  static_assert(I0 >= 0 && I1 <= LWin<LV>::n, "request range");
  if constexpr (!kLoaderTable<LV, CH>) {
    asm volatile("" : "+v"(pix_lane));  // keep the per-request arithmetic out of loop-invariant code motion (registers)
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      int c = ld + kLoaders * i;
      c = c < kChunks ? c : kChunks - 1;
      const unsigned idx = (unsigned)(c * kPPP) + pix_lane;
      const unsigned wy = idx / (unsigned)W::side, wx = idx - wy * (unsigned)W::side;
      const unsigned rel = __umul24(__umul24(wy, (unsigned)Wl) + wx, (unsigned)row_bytes) + lane_part;
      const int x = wx0 + (int)wx;
      const unsigned off = (!x_border || (unsigned)x < (unsigned)Wl) ? rel + (unsigned)tile_off : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(slab, (lptr_t)(win + c * 64), 16, (int)off, 0, 0, WM2F_DMA_AUX);
    }
    return;
  }
  if (!x_border) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      int c = ld + kLoaders * i;
      c = c < kChunks ? c : kChunks - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(slab, (lptr_t)(win + c * 64), 16, (int)(r.rel[i] + (unsigned)tile_off), 0, 0, WM2F_DMA_AUX);
    }
  } else {
    // the columns are tile-invariant too, but 77 more live registers do not fit: recompute them per use (the
    // empty asm hides the value from loop-invariant code motion)
    asm volatile("" : "+v"(pix_lane));
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      int c = ld + kLoaders * i;
      c = c < kChunks ? c : kChunks - 1;
      const unsigned idx = (unsigned)(c * kPPP) + pix_lane;
      const int x = wx0 + (int)(idx - (idx / (unsigned)W::side) * (unsigned)W::side);
      const unsigned off = ((unsigned)x < (unsigned)Wl) ? r.rel[i] + (unsigned)tile_off : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(slab, (lptr_t)(win + c * 64), 16, (int)off, 0, 0, WM2F_DMA_AUX);
    }
  }
}

struct LoaderTile {  // wave-uniform per-tile values of the loaders
  __amdgpu_buffer_rsrc_t slab[3];
  int tile_off[3], wx0[3];
  bool x_border;
};

__device__ __forceinline__ LoaderTile loader_tile(const float* value, const StreamGeom& sg, const TileId& t, int S, int heads,
                                                  int pixel_bytes = 128) {
  const QuadGeom& g = sg.q;
  const int row_stride = sg.v_pix, row_bytes = row_stride * 4;
  const float* vb = value + (int64_t)t.b * sg.v_img + (int64_t)t.h * sg.v_head + t.hh * 16;  // this head's slice (its second half: + 16 channels)
  LoaderTile lt;
  lt.x_border = false;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int Wl = g.W[l];
    const LevelTile lv = level_tile(g, l, t.tx, t.ty);
    lt.wx0[l] = lv.wx0;
    lt.tile_off[l] = (lv.wy0 * Wl + lv.wx0) * row_bytes;
    lt.x_border = lt.x_border || lv.wx0 < 0 || lv.wx0 + (l == 0 ? Win<0>::side : (l == 1 ? Win<1>::side : Win<2>::side)) > Wl;
    const int npx = g.W[l] * g.H[l];
    lt.slab[l] = __builtin_amdgcn_make_buffer_rsrc((void*)(vb + (int64_t)g.start[l] * row_stride), 0,
                                                   (npx - 1) * row_bytes + pixel_bytes, 0x00020000);
  }
  return lt;
}

// What a gather lane knows about "its" queries for one tile SHAPE (the numbers of queries per level in the tile: the
// same for every interior tile of an exact pyramid), so that a tile costs a handful of instructions per pass instead of a
// decode:
//   q_rel     row * W_lq + col of the query inside its own level, relative to the tile's first query of that level
//   lq        the query's level (0 coarse .. 2 fine);   col, row   its position among the tile's queries of that level
// With the tile's per-level first query (3 scalars) the token is q_rel + first[lq] and the query's column qx0[lq] + col;
// its reference point (qx + 0.5) / W_lq lands on pixel ref * W_l - 0.5 of level l (HF:993-1002 with grid_sample's
// align_corners = False rule), to which the sampling offset is added.
// General pyramids: packed into one register per pass (the kernel sits at its 168-register cap):
// col | row << 8 | lq << 16 | valid << 31.
struct PassConst {
  unsigned code[kPasses];
  __device__ __forceinline__ int col(int t) const { return (int)(code[t] & 255u); }
  __device__ __forceinline__ int row(int t) const { return (int)((code[t] >> 8) & 255u); }
  __device__ __forceinline__ int lq(int t) const { return (int)((code[t] >> 16) & 3u); }
  __device__ __forceinline__ bool valid(int t) const { return (code[t] >> 31) != 0u; }
};
// Exact 1 : 2 : 4 pyramids (the encoder's own shape; the kernel template is instantiated for them separately: the general
// arithmetic cost the production shape 15 %, profiles/r02_kbench_k1_general_vs_exact.jsonl): a query's reference point sits
// at pixel (tx * 4 * 2^l) + (cxs * 2^l - 0.5) of level l with cxs = (col + 0.5) * 2^-lq -- integers and eighths, exact in
// fp32, so the sampling coordinate `that + offset` is rounded once.
struct PassConstExact {
  int q_rel[kPasses], lq[kPasses];
  float cxs[kPasses], cys[kPasses];
  bool valid[kPasses];
};

// ---- flags instead of workgroup barriers (SYNC = 1) ---------------------------------------------------------
// With a barrier per phase all ten waves move in lock-step: everybody gathers (LDS saturated: the phases are bound
// by LDS bandwidth, 8 ds_read_b128 per point), then everybody computes coordinates / fetches / stores (LDS idle).
// Here each window has two `ready` counters (one per loader: "tiles loaded so far") and one `done` counter ("wave
// passes finished reading it"); a gather wave only waits for the window it is about to read, a loader only for the
// window it is about to overwrite.  Waves drift apart by up to a tile, so some gather while others do the rest.
// The polls are bounded (a lost wake-up ends in wrong results that the tests catch, never in a hung GPU).
constexpr int kCtrlReady = 0, kCtrlDone = 6, kCtrlWords = 12;
__device__ __forceinline__ void poll_ge(int* p, int target) {
  for (int it = 0; it < (1 << 18); ++it) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void publish(int* p, int v, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wave_done(int* p, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's window reads have returned
  if (lane == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// OPT: compile-time form of what were run-time switches in the kernel argument (bit 0: lane-major operand rows, bit 1: the
// round-1 loader schedule, bit 2: strip tile order).  As run-time fields they put both operand paths and both schedules
// into the one production kernel: +29 % instructions, 60 more scalar spills, 69 kernel-argument reloads inside the tile
// loop, and 152 -> 166 us on the model's operands (tools/probes/k1_two_libs.py, bisected over the round's commits).
template <bool FUSED, int MODE, int SYNC = 0, int CH = 8, bool EXACT = true, int OPT = 0>
__global__ __launch_bounds__(SCfg<CH>::THREADS, CH == 8 ? 1 : 4) void msdeform_stream_fwd_kernel(
    const float* __restrict__ value, const float* __restrict__ a_in, const float* __restrict__ b_in, float* __restrict__ out,
    StreamGeom sg, int S, int Q, int heads) {
  constexpr int D = 32, NL = 3, P = 4;
  constexpr bool kLanes = FUSED && (OPT & 1) != 0, kSched1 = (OPT & 2) == 0, kStrips = (OPT & 4) != 0;
  constexpr int kOrder = (OPT & 128) ? 3 : (OPT & 16) ? 2 : (kStrips ? 1 : 0);
  // OPT bit 7: SLAB order -- heads outermost.  An XCD's 32 workgroups walk the tiles of ONE (image, head) slab together
  // (2.75 MB of value at config 2: it stays in the XCD's 4 MiB L2, so the window halos -- 3.6 x the slab at L2 level -- are
  // fetched from HBM once) before the next head; needs head-major operand rows, otherwise a token's 1152-byte row is fetched
  // once per head.  Bits 8 / 9: non-temporal hint on the operand-row loads / the output stores (streams that must not evict
  // the slab).
  constexpr int kOpAux = (OPT & 256) ? 2 : WM2F_OP_AUX, kStAux = (OPT & 512) ? 2 : WM2F_ST_AUX;
  // OPT bit 10: the [offsets | logits] rows are bf16 (the merged projection's output under bf16 autocast, read as it is);
  // bit 11: the output is stored as bf16 (what the output projection reads there).  Training forward (wm2f_msdeform_rows_fwd).
  constexpr bool kRowsLp = FUSED && (OPT & 1024) != 0, kOutLp = (OPT & 2048) != 0;
  constexpr int kRowEsz = kRowsLp ? 2 : 4;
  static_assert(!(kRowsLp && (OPT & 1)), "bf16 rows exist in the token-major [offsets | logits] form only");
  // Lane rows: the 36 floats of a (token, head) record come in 16-byte ALIGNED pieces -- [x0 y0 x1 y1] of lanes 0..3, then
  // [x2 y2 w0 w1] of lanes 0..3, then w2 of lanes 0..3 (include/wm2f.h) -- so that a lane's two dwordx4 loads are 16-byte
  // aligned (nine consecutive floats per lane, a 36-byte lane stride, measured 1 % slower in the model).
  constexpr bool kAligned = kLanes;
  constexpr bool kAllFull = EXACT && (OPT & 8) != 0;  // level sides are multiples of the tile: every tile has the full query counts
  constexpr int kLoaderWave0 = SCfg<CH>::GW, kGW = SCfg<CH>::GW, kPB = SCfg<CH>::PB, kSplit = SCfg<CH>::SPLIT;
  static_assert(SYNC == 0 || CH == 8, "the flag-synchronised form exists for the full-head kernel only");
  __shared__ __attribute__((aligned(16))) float4 win0[SWin<0, CH>::chunks * 64];
  __shared__ __attribute__((aligned(16))) float4 win1[SWin<1, CH>::chunks * 64];
  __shared__ __attribute__((aligned(16))) float4 win2[SWin<2, CH>::chunks * 64];
  const QuadGeom& g = sg.q;
  // this workgroup's tiles: ids first + k * stride, k < n_my (XCD-contiguous ranges, as xcd_contiguous_id)
  const int xcd = blockIdx.x % kNumXcd, lw = blockIdx.x / kNumXcd;
  const int range_end = min((xcd + 1) * sg.per_xcd, sg.n_logical) - xcd * sg.per_xcd;  // ids of this XCD: [0, range_end)
  const int n_my = lw < range_end ? (range_end - lw + sg.wg_per_xcd - 1) / sg.wg_per_xcd : 0;
  const int first = xcd * sg.per_xcd + lw;
  if (n_my <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row_stride = sg.v_pix, row_bytes = row_stride * 4;
  __shared__ int ctrl[kCtrlWords];
  if (SYNC == 1) {
    if (tid < kCtrlWords) ctrl[tid] = 0;
    wg_barrier();
  }

  if (wave >= kLoaderWave0) {
    // ------------------------------------------------------------------ loader waves
    const int ld = wave - kLoaderWave0;
    // a request moves 1 KiB = 8 pixels x 128 B or 16 pixels x 64 B, lane-linear in LDS
    const unsigned pix_lane = CH == 8 ? (unsigned)lane >> 3 : (unsigned)lane >> 2;
    const unsigned lane_part = CH == 8 ? ((unsigned)lane & 7u) * 16u : ((unsigned)lane & 3u) * 16u;
    LoaderRegs<0> r0;
    LoaderRegs<1> r1;
    LoaderRegs<2> r2;
    if (kLoaderTable<0, CH>) loader_init<0, CH>(r0, ld, g.W[0], row_bytes, pix_lane, lane_part);
    if (kLoaderTable<1, CH>) loader_init<1, CH>(r1, ld, g.W[1], row_bytes, pix_lane, lane_part);
    loader_init<2, CH>(r2, ld, g.W[2], row_bytes, pix_lane, lane_part);
    // Schedule (per loader; F = fine window split in two parts so that no phase carries much more than a third of
    // a tile's requests -- the requests are accepted at the memory side's pace, ~100 cycles each per loader, and
    // the gather waves wait for the loader at every barrier):
    //   under the coarse gather of tile k:  F(k) part A                      (26 requests)
    //   under the mid gather:               F(k) part B, coarse(k + 1)       (17 + 13)
    //   under the fine gather:              [pause: the gather waves fetch their next operands]  mid(k + 1)  (21)
    constexpr int kFA = 26, kFB = LWin<2>::n - kFA;
    TileWalk walk = walk_init<kOrder>(first, sg, heads * kSplit);
    LoaderTile lt = loader_tile(value, sg, walk_tile(walk, sg, kSplit), S, heads, kPB);
    if (MODE != 5) loader_issue<0, 0, LWin<0>::n, CH>((lds4_t)win0, r0, lt.slab[0], ld, lt.tile_off[0], lt.x_border, lt.wx0[0], g.W[0], pix_lane, lane_part, row_bytes);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 5) loader_issue<1, 0, LWin<1>::n, CH>((lds4_t)win1, r1, lt.slab[1], ld, lt.tile_off[1], lt.x_border, lt.wx0[1], g.W[1], pix_lane, lane_part, row_bytes);
    if (SYNC == 1) {
      for (int k = 0; k < n_my; ++k) {
        wait_vm<LWin<1>::n>();  // coarse(k) landed
        publish(&ctrl[kCtrlReady + 0 * 2 + ld], k + 1, lane);
        wait_vm<0>();           // mid(k) landed
        publish(&ctrl[kCtrlReady + 1 * 2 + ld], k + 1, lane);
        poll_ge(&ctrl[kCtrlDone + 2], kGW * k);  // every gather wave is done with fine(k - 1)
        if (MODE != 5) loader_issue<2, 0, LWin<2>::n, CH>((lds4_t)win2, r2, lt.slab[2], ld, lt.tile_off[2], lt.x_border, lt.wx0[2], g.W[2], pix_lane);
        wait_vm<0>();
        publish(&ctrl[kCtrlReady + 2 * 2 + ld], k + 1, lane);
        if (k + 1 < n_my) {
          walk_step<kOrder>(walk, sg, heads * kSplit);
          lt = loader_tile(value, sg, walk_tile(walk, sg, kSplit), S, heads, kPB);
          poll_ge(&ctrl[kCtrlDone + 0], kGW * (k + 1));
          if (MODE != 5) loader_issue<0, 0, LWin<0>::n, CH>((lds4_t)win0, r0, lt.slab[0], ld, lt.tile_off[0], lt.x_border, lt.wx0[0], g.W[0], pix_lane, lane_part, row_bytes);
          __builtin_amdgcn_sched_barrier(0);
          poll_ge(&ctrl[kCtrlDone + 1], kGW * (k + 1));
          if (MODE != 5) loader_issue<1, 0, LWin<1>::n, CH>((lds4_t)win1, r1, lt.slab[1], ld, lt.tile_off[1], lt.x_border, lt.wx0[1], g.W[1], pix_lane, lane_part, row_bytes);
        }
      }
      return;
    }
    for (int k = 0; k < n_my; ++k) {
      const bool more = k + 1 < n_my;
      wait_vm<LWin<1>::n>();  // coarse(k) landed; mid(k) may still fly
      WM2F_SSTAMP(10, kLoaderWave0);
      wg_barrier();           // Bc(k): gather waves are done with fine(k-1)
      WM2F_SSTAMP(11, kLoaderWave0);
      if (MODE != 5) loader_issue<2, 0, kFA, CH>((lds4_t)win2, r2, lt.slab[2], ld, lt.tile_off[2], lt.x_border, lt.wx0[2], g.W[2], pix_lane);
      WM2F_SSTAMP(12, kLoaderWave0);
      wait_vm<kFA>();  // mid(k) landed
      wg_barrier();    // Bm(k): gather waves are done with coarse(k)
      WM2F_SSTAMP(13, kLoaderWave0);
      if (MODE != 5) loader_issue<2, kFA, LWin<2>::n, CH>((lds4_t)win2, r2, lt.slab[2], ld, lt.tile_off[2], lt.x_border, lt.wx0[2], g.W[2], pix_lane);
      if (kSched1) {
        // Per-wave stamps (profiles/r02_k1_stream_stamps_*.json): with coarse(k + 1) requested here the loaders were the
        // last to reach Bf(k) in every workgroup, 1.5k cycles behind the gather waves.  The coarse window is not needed
        // before Bc(k + 1), a whole fine gather away: request it behind Bf together with mid(k + 1).
        wait_vm<0>();  // fine(k) landed
        WM2F_SSTAMP(14, kLoaderWave0);
        wg_barrier();  // Bf(k): gather waves are done with mid(k)
        WM2F_SSTAMP(15, kLoaderWave0);
        if (more) {
          walk_step<kOrder>(walk, sg, heads * kSplit);
          lt = loader_tile(value, sg, walk_tile(walk, sg, kSplit), S, heads, kPB);
          __builtin_amdgcn_sched_barrier(0);
          if (MODE != 5) loader_issue<0, 0, LWin<0>::n, CH>((lds4_t)win0, r0, lt.slab[0], ld, lt.tile_off[0], lt.x_border, lt.wx0[0], g.W[0], pix_lane, lane_part, row_bytes);
          __builtin_amdgcn_sched_barrier(0);
          if (MODE != 5) loader_issue<1, 0, LWin<1>::n, CH>((lds4_t)win1, r1, lt.slab[1], ld, lt.tile_off[1], lt.x_border, lt.wx0[1], g.W[1], pix_lane, lane_part, row_bytes);
        }
        continue;
      }
      if (more) {
        walk_step<kOrder>(walk, sg, heads * kSplit);
        lt = loader_tile(value, sg, walk_tile(walk, sg, kSplit), S, heads, kPB);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 5) loader_issue<0, 0, LWin<0>::n, CH>((lds4_t)win0, r0, lt.slab[0], ld, lt.tile_off[0], lt.x_border, lt.wx0[0], g.W[0], pix_lane, lane_part, row_bytes);
        wait_vm<LWin<0>::n>();  // fine(k) landed
      } else {
        wait_vm<0>();
      }
      WM2F_SSTAMP(14, kLoaderWave0);
      wg_barrier();  // Bf(k): gather waves are done with mid(k)
      WM2F_SSTAMP(15, kLoaderWave0);
      if (more) {
        __builtin_amdgcn_s_sleep(24);  // ~1.5k cycles: leave the memory path to the gather waves' operand loads
        if (MODE != 5) loader_issue<1, 0, LWin<1>::n, CH>((lds4_t)win1, r1, lt.slab[1], ld, lt.tile_off[1], lt.x_border, lt.wx0[1], g.W[1], pix_lane, lane_part, row_bytes);
      }
    }
    (void)kFB;
    return;
  }

  // -------------------------------------------------------------------- gather waves
  // OPT bit 5 (A/B): the second-dispatched half of the gather waves (4 - 7: the loser of every issue arbitration with its SIMD's
  // older wave, and the waves the per-wave stamps show on the critical path) at static priority 1
  if ((OPT & 32) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if ((OPT & 64) && wave < 4) __builtin_amdgcn_s_setprio(1);  // the opposite choice
  const int j = tid & 3, quad = lane >> 2;
  const int xq = (0x73261540 >> ((quad & 7) * 4)) & 7;  // bank-aware quad -> query order, see the kernel above
  const int slot = wave * 16 + (quad & 8) + xq;
  // CH = 8: a lane owns 4 + 4 channels, the quad's two 64-byte halves of a 128-byte pixel (hq: which half first, for the
  // banks).  CH = 4: a lane owns 4 channels, the quad covers the 64-byte pixel; the four quads of an LDS 16-lane group are
  // x-neighbours (the xq order above), so for a slowly varying offset field they read four consecutive 64-byte pixels
  // = all 64 banks once.
  const int hq = CH == 8 ? (quad >> 2) & 1 : 0;
  const int off1 = j * 16 + hq * 64, off2 = j * 16 + (1 - hq) * 64;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_in, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)b_in, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);

  typename std::conditional<EXACT, PassConstExact, PassConst>::type pc;
  int shape_key = -1;  // packed (nqx, nqy) per level of the tile shape `pc` was built for
  // operands of a tile: raw loads + where they go
  struct Ops {
    float2 lc[kPasses][NL];
    float wt[kPasses][NL];
    int qrow[kPasses];  // b * Q + token
    bool valid[kPasses];
    int wx0[NL], wy0[NL], qx0[NL], qy0[NL], b, h, tx, ty, hh;
  };
  // reference point of token q (HF:1127-1156): ((column + 0.5) / W_l, (row + 0.5) / H_l) of its own level (slow path only)
  auto ref_point = [&](int q, float& rx, float& ry) __attribute__((always_inline)) {
    const int l = (q >= g.start[1] ? 1 : 0) + (q >= g.start[2] ? 1 : 0);
    const int rel = q - (l == 2 ? g.start[2] : (l == 1 ? g.start[1] : 0));
    const int Wq = l == 2 ? g.W[2] : (l == 1 ? g.W[1] : g.W[0]);
    const float iw = l == 2 ? g.inv_w[2] : (l == 1 ? g.inv_w[1] : g.inv_w[0]), ih = l == 2 ? g.inv_h[2] : (l == 1 ? g.inv_h[1] : g.inv_h[0]);
    const int qy = (int)(((float)rel + 0.5f) * iw), r0 = rel - qy * Wq;
    const int qyc = r0 < 0 ? qy - 1 : (r0 >= Wq ? qy + 1 : qy);  // one correction step: rel can exceed 2^22 / W
    const int qx = rel - qyc * Wq;
    rx = ((float)qx + 0.5f) * iw;
    ry = ((float)qyc + 0.5f) * ih;
  };
  const int a_row = g.a_qstride * kRowEsz, b_row = g.b_qstride * kRowEsz;  // bytes per token in the two operand arrays (< 2^24: host)
  auto fetch = [&](const TileId& t) __attribute__((always_inline)) {
    Ops o;
    o.b = t.b;
    o.h = t.h;
    o.tx = t.tx;
    o.ty = t.ty;
    o.hh = t.hh;
    int nqx[NL], nqy[NL], qfirst[NL], key = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if constexpr (EXACT) {
        // (window origins are tx * fq - 1 - margin: recomputed where they are used instead of carried in `Ops` -- the
        // kernel's scalar registers are its scarcest resource: every per-tile scalar spilled is a v_readlane or, worse, a
        // kernel-argument reload with its wait on the tile's critical path)
        const int Wl = g.W0 << l, Hl = g.H0 << l, fq = kQF >> (2 - l);
        if (kAllFull) {  // level sides are multiples of the tile: every tile holds fq x fq queries of level l
          nqx[l] = nqy[l] = fq;
        } else {
          int nx = Wl - t.tx * fq, ny = Hl - t.ty * fq;
          nqx[l] = nx < 0 ? 0 : (nx > fq ? fq : nx);
          nqy[l] = ny < 0 ? 0 : (ny > fq ? fq : ny);
        }
        qfirst[l] = t.b * Q + g.start[l] + (t.ty * fq) * Wl + t.tx * fq;  // the tile's first token of level l
      } else {
        const LevelTile lv = level_tile(g, l, t.tx, t.ty);
        o.wx0[l] = lv.wx0;
        o.wy0[l] = lv.wy0;
        o.qx0[l] = lv.qx0;
        o.qy0[l] = lv.qy0;
        nqx[l] = lv.nqx;
        nqy[l] = lv.nqy;
        qfirst[l] = t.b * Q + g.start[l] + lv.qy0 * g.W[l] + lv.qx0;
      }
      key = key * 1024 + nqx[l] * 32 + nqy[l];
    }
    if (key != shape_key) {  // wave-uniform; interior tiles all share one shape
      shape_key = key;
      const int c1 = nqx[0] * nqy[0], c2 = c1 + nqx[1] * nqy[1], nq = c2 + nqx[2] * nqy[2];
#pragma unroll
      for (int t2 = 0; t2 < kPasses; ++t2) {
        // passes 0, 1: all 8 gather waves; the third pass of a full tile (80 queries) goes to waves 0-3 and 7:
        // the loaders share their SIMDs with waves (0, 4) and (1, 5) (waves go to SIMDs cyclically), so waves 4
        // and 5 -- and 6, whose SIMD then carries 3 + 2 passes -- stay at two passes
        // With 10 gather waves two passes cover 320 queries and wave 2 takes the last 16.
        // Half-head form: 7 gather waves x 16 quads x 3 passes = 336 = a full tile's queries: every wave takes three passes.
        const int w3 = kGW == 7 ? wave : (kGW == 8 ? (wave < 4 ? wave : (wave == 7 ? 4 : 1 << 20)) : (wave == 2 ? 0 : 1 << 20));
        int qi = t2 < 2 ? slot + (kGW * 16) * t2 : 2 * (kGW * 16) + w3 * 16 + (quad & 8) + xq;
        const bool ok = qi < nq;
        if (!ok) qi = 0;  // an empty slot shadows the tile's first query (a real token: loads stay in range)
        const bool ge1 = qi >= c1, ge2 = qi >= c2;
        const int lq = (ge1 ? 1 : 0) + (ge2 ? 1 : 0);
        const int nx = ge2 ? nqx[2] : (ge1 ? nqx[1] : nqx[0]);
        const int loc_i = qi - (ge2 ? c2 : (ge1 ? c1 : 0));
        const int nxs = nx < 1 ? 1 : nx;
        const int ly_ = (int)(((float)loc_i + 0.5f) * __builtin_amdgcn_rcpf((float)nxs));  // exact: small integers
        const int lx_ = loc_i - ly_ * nxs;
        if constexpr (EXACT) {
          const float sc = ge2 ? 0.25f : (ge1 ? 0.5f : 1.f);
          pc.valid[t2] = ok;
          pc.lq[t2] = lq;
          pc.q_rel[t2] = (int)__umul24((unsigned)ly_, (unsigned)(g.W0 << lq)) + lx_;
          pc.cxs[t2] = ((float)lx_ + 0.5f) * sc;
          pc.cys[t2] = ((float)ly_ + 0.5f) * sc;
        } else {
          pc.code[t2] = (unsigned)lx_ | ((unsigned)ly_ << 8) | ((unsigned)lq << 16) | (ok ? 0x80000000u : 0u);
        }
      }
    }
#pragma unroll
    for (int t2 = 0; t2 < kPasses; ++t2) {
      int q;
      if constexpr (EXACT) {
        q = pc.q_rel[t2] + (pc.lq[t2] == 2 ? qfirst[2] : (pc.lq[t2] == 1 ? qfirst[1] : qfirst[0]));
        o.valid[t2] = pc.valid[t2];
      } else {
        const int lq = pc.lq(t2);
        q = (int)__umul24((unsigned)pc.row(t2), (unsigned)(lq == 2 ? g.W[2] : (lq == 1 ? g.W[1] : g.W[0]))) + pc.col(t2) +
            (lq == 2 ? qfirst[2] : (lq == 1 ? qfirst[1] : qfirst[0]));
        o.valid[t2] = pc.valid(t2);
      }
      o.qrow[t2] = q;
    }
    const int ah = (o.h * (NL * P * 2) + j * 2) * kRowEsz, bh = (o.h * (NL * P) + j) * kRowEsz;
#pragma unroll
    for (int t2 = 0; t2 < kPasses; ++t2) {
      int q = o.qrow[t2];
      if (MODE == 8) q = o.qrow[0] & ~1023;  // (profiling build, timing ablation: every operand load from one hot record -- same instructions, L2 latency)
      const int a_off = (int)__umul24((unsigned)q, (unsigned)a_row) + ah;
      const int b_off = (int)__umul24((unsigned)q, (unsigned)b_row) + bh;
      if (kLanes) {
        // lane rows (wm2f_msdeform_fused_lanes_fwd): the 36 numbers of head h form one 144-byte record -- a pass is 3 loads
        // (two 16-byte aligned dwordx4, one dword) whose quad footprint lies inside that record, instead of 6 loads
        // scattered over the token's 1152-byte row (16 quads x 6 loads x 8 waves queued
        // in the texture-address unit behind the loaders' traffic: the per-wave stamps showed 3.0k cycles for the older and
        // 5.5k for the younger wave of each SIMD in this fetch)
        // b_row is the HEAD stride of the lane-major array here (144 B inside a 1152-B token row, or a whole (B, Q, 36)
        // slab when the rows are stored head-major)
        const int rec = (int)__umul24((unsigned)q, (unsigned)a_row) + o.h * b_row;
        const int off = rec + (kAligned ? j * 16 : j * 36);
        f32x4q A = {0.f, 0.f, 0.f, 0.f}, Bq = {0.f, 0.f, 0.f, 0.f};
        if (MODE != 6) {  // (MODE 6, profiling build: timing ablation without the gather waves' loads and stores)
          A = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off, 0, kOpAux));
          Bq = __builtin_bit_cast(f32x4q, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off + (kAligned ? 64 : 16), 0, kOpAux));
        }
        o.lc[t2][0] = make_float2(A.x, A.y);
        o.lc[t2][1] = make_float2(A.z, A.w);
        o.lc[t2][2] = make_float2(Bq.x, Bq.y);
        o.wt[t2][0] = Bq.z;
        o.wt[t2][1] = Bq.w;
        o.wt[t2][2] = MODE == 6 ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rs, kAligned ? rec + 128 + j * 4 : off + 32, 0, kOpAux));
      } else {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          if constexpr (kRowsLp) {  // bf16 -> fp32 is a shift: exact
            const unsigned xy = __builtin_amdgcn_raw_buffer_load_b32(a_rs, a_off + l * (P * 2 * 2), 0, kOpAux);
            const unsigned lg = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(b_rs, b_off + l * (P * 2), 0, kOpAux);
            o.lc[t2][l] = make_float2(__uint_as_float(xy << 16), __uint_as_float(xy & 0xffff0000u));
            o.wt[t2][l] = __uint_as_float(lg << 16);
          } else {
            o.lc[t2][l] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(a_rs, a_off + l * (P * 2 * 4), 0, kOpAux));
            o.wt[t2][l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rs, b_off + l * (P * 4), 0, kOpAux));
          }
        }
      }
    }
    return o;
  };

  TileWalk walk = walk_init<kOrder>(first, sg, heads * kSplit);
  Ops nxt = fetch(walk_tile(walk, sg, kSplit));
  for (int k = 0; k < n_my; ++k) {
    Ops cur = nxt;
    WM2F_SSTAMP(0, 0);
    // ---- per-point pixel coordinates and (fused) softmax weights
    float px[kPasses][NL], py[kPasses][NL], wt[kPasses][NL];
#pragma unroll
    for (int t = 0; t < kPasses; ++t) {
#pragma unroll
      for (int l = 0; l < NL; ++l) wt[t][l] = cur.wt[t][l];
      if (FUSED) {  // softmax over the 12 logits of the quad (HF:986-991)
        const float mx = quad_max(fmaxf(fmaxf(wt[t][0], wt[t][1]), wt[t][2]));
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          wt[t][l] = __expf(wt[t][l] - mx);
          s += wt[t][l];
        }
        const float inv = __builtin_amdgcn_rcpf(quad_sum(s));
#pragma unroll
        for (int l = 0; l < NL; ++l) wt[t][l] *= inv;
      }
      float refx = 0.f, refy = 0.f;
      if constexpr (FUSED && !EXACT) {  // the query's reference point (`pc` still describes this tile: the next fetch comes later)
        const int lq = pc.lq(t);
        const int qx = pc.col(t) + (lq == 2 ? cur.qx0[2] : (lq == 1 ? cur.qx0[1] : cur.qx0[0]));
        const int qy = pc.row(t) + (lq == 2 ? cur.qy0[2] : (lq == 1 ? cur.qy0[1] : cur.qy0[0]));
        refx = ((float)qx + 0.5f) * (lq == 2 ? g.inv_w[2] : (lq == 1 ? g.inv_w[1] : g.inv_w[0]));
        refy = ((float)qy + 0.5f) * (lq == 2 ? g.inv_h[2] : (lq == 1 ? g.inv_h[1] : g.inv_h[0]));
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        if constexpr (FUSED && EXACT) {
          // loc = ref + off / (W, H); pixel = loc * (W, H) - 0.5 == ref * W - 0.5 + off, with ref * W - 0.5 taken
          // exactly: tile origin + (cxs * 2^l - 0.5)
          px[t][l] = ((float)(cur.tx << (l + 2)) + fmaf(pc.cxs[t], (float)(1 << l), -0.5f)) + cur.lc[t][l].x;
          py[t][l] = ((float)(cur.ty << (l + 2)) + fmaf(pc.cys[t], (float)(1 << l), -0.5f)) + cur.lc[t][l].y;
        } else if constexpr (FUSED) {
          px[t][l] = fmaf(refx, (float)g.W[l], -0.5f) + cur.lc[t][l].x;
          py[t][l] = fmaf(refy, (float)g.H[l], -0.5f) + cur.lc[t][l].y;
        } else {
          const float Wl = (float)g.W[l], Hl = (float)g.H[l];
          px[t][l] = ((2.f * cur.lc[t][l].x - 1.f + 1.f) * Wl - 1.f) * 0.5f;
          py[t][l] = ((2.f * cur.lc[t][l].y - 1.f + 1.f) * Hl - 1.f) * 0.5f;
        }
      }
    }
    Acc acc[kPasses];
    unsigned slow[kPasses];
#pragma unroll
    for (int t = 0; t < kPasses; ++t) {
      acc[t].a_lo = acc[t].a_hi = acc[t].b_lo = acc[t].b_hi = (f32x2){0.f, 0.f};
      slow[t] = 0;
    }
    const bool skip_last = __builtin_amdgcn_ballot_w64(cur.valid[kPasses - 1]) == 0;
    auto wx0_of = [&](int l) __attribute__((always_inline)) { return EXACT ? cur.tx * (kQF >> (2 - l)) - 1 - kQM : cur.wx0[l]; };
    auto wy0_of = [&](int l) __attribute__((always_inline)) { return EXACT ? cur.ty * (kQF >> (2 - l)) - 1 - kQM : cur.wy0[l]; };

    auto window_ready = [&](int w) __attribute__((always_inline)) {
      if (SYNC == 1) {
        poll_ge(&ctrl[kCtrlReady + w * 2 + 0], k + 1);
        poll_ge(&ctrl[kCtrlReady + w * 2 + 1], k + 1);
      } else {
        wg_barrier();
      }
    };
    WM2F_SSTAMP(1, 0);
    window_ready(0);  // Bc(k)
    WM2F_SSTAMP(2, 0);
    gather_phase<0, MODE, true, CH>(win0, acc, px, py, wt, cur.valid, wx0_of(0), wy0_of(0), slow, off1, off2, skip_last);
    if (SYNC == 1) wave_done(&ctrl[kCtrlDone + 0], lane);
    WM2F_SSTAMP(3, 0);
    window_ready(1);  // Bm(k)
    WM2F_SSTAMP(4, 0);
    gather_phase<1, MODE, true, CH>(win1, acc, px, py, wt, cur.valid, wx0_of(1), wy0_of(1), slow, off1, off2, skip_last);
    if (SYNC == 1) wave_done(&ctrl[kCtrlDone + 1], lane);
    WM2F_SSTAMP(5, 0);
    window_ready(2);  // Bf(k)
    WM2F_SSTAMP(6, 0);
    if (CH == 8 && k + 1 < n_my) {  // lands under the fine gather
      walk_step<kOrder>(walk, sg, heads * kSplit);
      nxt = fetch(walk_tile(walk, sg, kSplit));
    }
    __builtin_amdgcn_sched_barrier(0);
    WM2F_SSTAMP(7, 0);
    gather_phase<2, MODE, true, CH>(win2, acc, px, py, wt, cur.valid, wx0_of(2), wy0_of(2), slow, off1, off2, skip_last);
    if (SYNC == 1) wave_done(&ctrl[kCtrlDone + 2], lane);
    WM2F_SSTAMP(8, 0);
    if (CH == 4 && k + 1 < n_my) {
      // half-head form: 128 registers per wave leave no room to hold the next tile's 27 operand registers through the
      // fine gather; they are requested here and land under the stores, the loop turn and the other workgroup's work
      __builtin_amdgcn_sched_barrier(0);
      walk_step<kOrder>(walk, sg, heads * kSplit);
      nxt = fetch(walk_tile(walk, sg, kSplit));
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- slow points (rare), then the stores
    const float* vb = value + (int64_t)cur.b * sg.v_img + (int64_t)cur.h * sg.v_head + cur.hh * 16;
    const bool wave_slow = __builtin_amdgcn_ballot_w64((slow[0] | slow[1] | slow[2]) != 0) != 0;
#pragma unroll
    for (int t = 0; t < kPasses; ++t) {
      float4 r1 = make_float4(acc[t].a_lo.x, acc[t].a_lo.y, acc[t].a_hi.x, acc[t].a_hi.y);
      float4 r2 = make_float4(acc[t].b_lo.x, acc[t].b_lo.y, acc[t].b_hi.x, acc[t].b_hi.y);
      if (wave_slow && cur.valid[t]) {
        unsigned todo = (unsigned)bcast<0>((int)slow[t]) | ((unsigned)bcast<1>((int)slow[t]) << 3) |
                        ((unsigned)bcast<2>((int)slow[t]) << 6) | ((unsigned)bcast<3>((int)slow[t]) << 9);
        constexpr bool lm = kLanes;  // lane-major rows: point k2 of level l sits at [k2 * 9 + 2 * l], its logit at [k2 * 9 + 6 + l]
        // (element offsets: the rows are fp32, or bf16 in the training form)
        const int64_t ap_e = lm ? (int64_t)cur.qrow[t] * g.a_qstride + (int64_t)cur.h * g.b_qstride
                                : (int64_t)cur.qrow[t] * g.a_qstride + cur.h * (NL * P * 2);
        const int64_t bp_e = lm ? ap_e : (int64_t)cur.qrow[t] * g.b_qstride + cur.h * (NL * P);
        const float* b_base = lm ? a_in : b_in;
        auto row_a = [&](int i) __attribute__((always_inline)) {
          if constexpr (kRowsLp) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a_in)[ap_e + i] << 16);
          else return a_in[ap_e + i];
        };
        auto row_b = [&](int i) __attribute__((always_inline)) {
          if constexpr (kRowsLp) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(b_base)[bp_e + i] << 16);
          else return b_base[bp_e + i];
        };
        // where point k2 of level l keeps its x (y follows) and its logit inside the 36-float record
        auto xy_at = [&](int k2, int l) __attribute__((always_inline)) {
          return kAligned ? (l < 2 ? k2 * 4 + 2 * l : 16 + k2 * 4) : (lm ? k2 * 9 + 2 * l : (l * P + k2) * 2);
        };
        auto logit_at = [&](int i) __attribute__((always_inline)) {  // i = l * P + k2
          const int k2 = i & 3, l = i >> 2;
          return kAligned ? row_b(l < 2 ? 16 + k2 * 4 + 2 + l : 32 + k2) : (lm ? row_b(k2 * 9 + 6 + l) : row_b(i));
        };
        float refx = 0.f, refy = 0.f, sm_max = 0.f, sm_inv = 1.f;
        if (FUSED && todo) {  // re-derive what the fast path no longer holds in registers
          ref_point(cur.qrow[t] - cur.b * Q, refx, refy);
          sm_max = logit_at(0);
          for (int i = 1; i < NL * P; ++i) sm_max = fmaxf(sm_max, logit_at(i));
          float ssum = 0.f;
          for (int i = 0; i < NL * P; ++i) ssum += __expf(logit_at(i) - sm_max);
          sm_inv = __builtin_amdgcn_rcpf(ssum);
        }
        while (todo) {  // bit (k2 * 3 + l): point k2 of level l
          const int i = __ffs(todo) - 1;
          todo &= todo - 1;
          const int k2 = i / 3, l = i - k2 * 3;
          const int Wl = l == 2 ? g.W[2] : (l == 1 ? g.W[1] : g.W[0]), Hl = l == 2 ? g.H[2] : (l == 1 ? g.H[1] : g.H[0]);
          const int st_l = l == 0 ? g.start[0] : (l == 1 ? g.start[1] : g.start[2]);
          const float lx = row_a(xy_at(k2, l)), ly = row_a(xy_at(k2, l) + 1);
          float aw = logit_at(l * P + k2), x, y;
          if (FUSED) {
            aw = __expf(aw - sm_max) * sm_inv;
            x = (refx * (float)Wl - 0.5f) + lx;
            y = (refy * (float)Hl - 0.5f) + ly;
          } else {
            x = ((2.f * lx - 1.f + 1.f) * (float)Wl - 1.f) * 0.5f;
            y = ((2.f * ly - 1.f + 1.f) * (float)Hl - 1.f) * 0.5f;
          }
          const float* vlev = vb + (int64_t)st_l * row_stride;
          quad_point_slow(r1, vlev + (off1 >> 2), Hl, Wl, row_stride, x, y, aw);
          if (CH == 8) quad_point_slow(r2, vlev + (off2 >> 2), Hl, Wl, row_stride, x, y, aw);
        }
      }
      const unsigned o_off = cur.valid[t] ? (unsigned)((cur.qrow[t] * heads + cur.h) * (D * (kOutLp ? 2 : 4)) + cur.hh * (kOutLp ? 32 : 64)) : kOobOffset;
      if (MODE == 6) {  // keep the sums alive without a store
        asm volatile("" ::"v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w), "v"(r2.x), "v"(r2.y), "v"(r2.z), "v"(r2.w));
        continue;
      }
      if constexpr (kOutLp) {  // round to nearest even, as a cast pass over the fp32 output would
        __builtin_amdgcn_raw_buffer_store_b64((u32x2q){pack_bf16_rne(r1.x, r1.y), pack_bf16_rne(r1.z, r1.w)}, out_rs, (int)(o_off + (off1 >> 1)), 0, kStAux);
        if (CH == 8) __builtin_amdgcn_raw_buffer_store_b64((u32x2q){pack_bf16_rne(r2.x, r2.y), pack_bf16_rne(r2.z, r2.w)}, out_rs, (int)(o_off + (off2 >> 1)), 0, kStAux);
        continue;
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r1), out_rs, (int)(o_off + off1), 0, kStAux);
      if (CH == 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r2), out_rs, (int)(o_off + off2), 0, kStAux);
    }
    WM2F_SSTAMP(9, 0);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------ host side
#ifdef WM2F_PROFILING
// Applies when: 3 levels ordered coarse -> fine with sides exactly 1 : 2 : 4, 4 points, queries == tokens.
template <bool FUSED>
int launch_quad(const void* value, const void* a, const void* b, void* out, const int32_t* level_hw, int B, int S, int Q,
                int heads, int L, int P, void* stream, const char* who, bool* handled, int mode, int a_qstride,
                int b_qstride) {
  *handled = false;
  if (P != 4 || L != 3 || (int64_t)Q != S) return WM2F_OK;
  const int H0 = level_hw[0], W0 = level_hw[1];
  for (int l = 1; l < 3; ++l)
    if (level_hw[2 * l] != (H0 << l) || level_hw[2 * l + 1] != (W0 << l)) return WM2F_OK;
  if (H0 < 1 || W0 < 1 || (int64_t)H0 * W0 * 21 != S) return WM2F_OK;
  QuadGeom g;
  g.W0 = W0;
  g.H0 = H0;
  g.tiles_x = (4 * W0 + kQF - 1) / kQF;
  g.tiles_y = (4 * H0 + kQF - 1) / kQF;
  g.start[0] = 0;
  g.start[1] = H0 * W0;
  g.start[2] = 5 * H0 * W0;
  g.a_qstride = a_qstride > 0 ? a_qstride : heads * L * P * 2;
  g.b_qstride = b_qstride > 0 ? b_qstride : heads * L * P;
  const int64_t n_logical = (int64_t)B * heads * g.tiles_x * g.tiles_y;
  if (n_logical > (1 << 30)) return WM2F_OK;
  // 32-bit byte offsets into loc / weights / out, 24-bit multiplies in the window addressing
  const int64_t lim = 0x7fffffff;
  if ((int64_t)B * Q * g.a_qstride * 4 >= lim || (int64_t)B * Q * g.b_qstride * 4 >= lim ||
      (int64_t)B * Q * heads * 32 * 4 >= lim || (int64_t)16 * H0 * W0 >= (1 << 24) || heads * 32 * 4 >= (1 << 24))
    return WM2F_OK;
  const int per_xcd = (int)ceil_div64(n_logical, kNumXcd);
  auto kfn = msdeform_quad_fwd_kernel<FUSED, 0>;
#ifdef WM2F_PROFILING
  if (mode == 1) kfn = msdeform_quad_fwd_kernel<FUSED, 1>;
  if (mode == 2) kfn = msdeform_quad_fwd_kernel<FUSED, 2>;
  if (mode == 4) kfn = msdeform_quad_fwd_kernel<FUSED, 4>;
  if (mode == 7) kfn = msdeform_quad_fwd_kernel<FUSED, 7>;
#endif
  hipLaunchKernelGGL(kfn, dim3(per_xcd * kNumXcd), dim3(kThreads), 0, (hipStream_t)stream, (const float*)value,
                     (const float*)a, (const float*)b, (float*)out, g, S, Q, heads, (int)n_logical, per_xcd);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: quad launch failed: %s", who, hipGetErrorString(e));
    return WM2F_ELAUNCH;
  }
  *handled = true;
  return WM2F_OK;
}

template int launch_quad<false>(const void*, const void*, const void*, void*, const int32_t*, int, int, int, int, int,
                                int, void*, const char*, bool*, int, int, int);
template int launch_quad<true>(const void*, const void*, const void*, void*, const int32_t*, int, int, int, int, int,
                               int, void*, const char*, bool*, int, int, int);


#endif  // WM2F_PROFILING (launch_quad)

// Streaming launch: one workgroup per CU.
template <bool FUSED>
int launch_stream(const void* value, const void* a, const void* b, void* out, const int32_t* level_hw, int B, int S,
                  int Q, int heads, int L, int P, void* stream, const char* who, bool* handled, int mode, int a_qstride,
                  int b_qstride, int lanes) {
  // mode: 0 the kernel that runs: full-head form (one workgroup per CU), raster tile order, coarse(k + 1) requested behind Bf;
  //       400 the half-head form (two workgroups per CU; measured SLOWER: see SCfg); 100 flags instead of barriers;
  //       200 tiles in 2-wide vertical strips; 300 the round-1 loader schedule;
  //       4 / 7 (profiling build) without LDS reads / stamped; 74 half-head stamped
  const bool half = (mode == 400 || mode == 74);
  const int split = half ? 2 : 1;
  *handled = false;
  if (P != 4 || L != 3 || (int64_t)Q != S) return WM2F_OK;
  const int H0 = level_hw[0], W0 = level_hw[1];
  bool exact = true;
  int64_t tokens = 0;
  for (int l = 0; l < 3; ++l) {
    const int Hl = level_hw[2 * l], Wl = level_hw[2 * l + 1];
    if (Hl < 1 || Wl < 1) return WM2F_OK;
    if (Hl != (H0 << l) || Wl != (W0 << l)) exact = false;
    tokens += (int64_t)Hl * Wl;
  }
  if (tokens != S) return WM2F_OK;
  if (!exact) {
    // Any other coarse -> fine pyramid whose levels roughly double (input sizes that are not multiples of 32: 25x42 /
    // 50x84 / 100x167): a 16 x 16 tile of the finest level must hold at most 4+1, 8+1 queries per axis of the two coarser
    // levels (the slot count: 25 + 81 + 256 <= 384) and its sampling range must fit the compile-time windows; the float
    // divisions of the tile -> query arithmetic must stay exact (operands below 2^22).  Full-head form only.
    if (half || lanes || mode == 100 || mode == 4 || mode == 7 || mode == 74) return WM2F_OK;
    const int Wf = level_hw[5], Hf = level_hw[4];
    for (int l = 0; l < 2; ++l) {
      const int Hl = level_hw[2 * l], Wl = level_hw[2 * l + 1], fq = kQF >> (2 - l);
      if ((int64_t)kQF * Wl > (int64_t)(fq + 1) * Wf - Wl || (int64_t)kQF * Hl > (int64_t)(fq + 1) * Hf - Hl) return WM2F_OK;  // <= fq + 1 queries per axis
      if ((int64_t)kQF * Wl > (int64_t)fq * Wf + Wf || (int64_t)kQF * Hl > (int64_t)fq * Hf + Hf) return WM2F_OK;          // window: 16 W_l / W_f <= fq + 1
      if ((int64_t)2 * kQF * ((Wf + kQF - 1) / kQF + 1) * Wl + 2 * Wf >= (1 << 22) ||
          (int64_t)2 * kQF * ((Hf + kQF - 1) / kQF + 1) * Hl + 2 * Hf >= (1 << 22))
        return WM2F_OK;
    }
  }
  StreamGeom sg;
  QuadGeom& g = sg.q;
  g.W0 = W0;
  g.H0 = H0;
  g.exact = exact ? 1 : 0;
  int64_t start = 0;
  for (int l = 0; l < 3; ++l) {
    g.H[l] = level_hw[2 * l];
    g.W[l] = level_hw[2 * l + 1];
    g.inv_w[l] = 1.f / (float)g.W[l];
    g.inv_h[l] = 1.f / (float)g.H[l];
    g.start[l] = (int)start;
    start += (int64_t)g.H[l] * g.W[l];
  }
  g.inv_2wf = 1.f / (float)(2 * g.W[2]);
  g.inv_2hf = 1.f / (float)(2 * g.H[2]);
  g.tiles_x = (g.W[2] + kQF - 1) / kQF;
  g.tiles_y = (g.H[2] + kQF - 1) / kQF;
  g.a_qstride = a_qstride > 0 ? a_qstride : heads * L * P * 2;
  g.b_qstride = b_qstride > 0 ? b_qstride : heads * L * P;
  const int64_t n_logical = (int64_t)B * heads * split * g.tiles_x * g.tiles_y;
  const int64_t lim = 0x7fffffff;
  if (n_logical >= (1 << 22) || (int64_t)B * Q >= (1 << 24) || g.a_qstride * 4 >= (1 << 24) || (!(lanes & 1) && g.b_qstride * 4 >= (1 << 24)) || ((lanes & 1) && (int64_t)heads * g.b_qstride * 4 >= lim) ||
      (int64_t)B * Q * g.a_qstride * 4 >= lim || (!(lanes & 1) && (int64_t)B * Q * g.b_qstride * 4 >= lim) ||
      (int64_t)B * Q * heads * 32 * 4 >= lim || (int64_t)g.H[2] * g.W[2] >= (1 << 24) || heads * 32 * 4 >= (1 << 24) ||
      (int64_t)S * heads * 32 * 4 >= lim)
    return WM2F_OK;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("%s: cannot query the device", who);
      return WM2F_ELAUNCH;
    }
    n_cu = prop.multiProcessorCount;
  }
  int wg = (n_cu - n_cu % kNumXcd) * split;  // a multiple of the XCD count; the half-head form keeps two workgroups per CU
  if (wg < kNumXcd) wg = kNumXcd;
  sg.n_logical = (int)n_logical;
  sg.per_xcd = (int)ceil_div64(n_logical, kNumXcd);
  sg.wg_per_xcd = wg / kNumXcd;
  if (sg.wg_per_xcd > sg.per_xcd) sg.wg_per_xcd = sg.per_xcd;
  wg = sg.wg_per_xcd * kNumXcd;
  sg.inv_heads = 1.f / (float)(heads * split);
  sg.inv_ntiles = 1.f / (float)(g.tiles_x * g.tiles_y);
  sg.inv_tiles_x = 1.f / (float)g.tiles_x;
  sg.step_t = sg.wg_per_xcd / (heads * split);
  sg.step_h = sg.wg_per_xcd % (heads * split);
  sg.step_tx = sg.step_ty = 0;
  // slab order (lanes bit 2; profiling build: modes 800-803, bit 0 / 1 = non-temporal operand loads / output stores)
  const bool slab = exact && FUSED && (lanes & 1) && !half && ((lanes & 4) || (mode >= 800 && mode <= 820));
  if (slab) {
    const int n_tiles = g.tiles_x * g.tiles_y;
    sg.step_h = sg.wg_per_xcd / n_tiles;
    sg.step_t = sg.wg_per_xcd % n_tiles;
    sg.step_ty = sg.step_t / g.tiles_x;
    sg.step_tx = sg.step_t % g.tiles_x;
  }
  // tile work order inside an image: plain raster; mode 200 = 2-wide vertical strips (see StreamGeom; A/B measurement:
  // 169 against 165 us -- the seams a strip order saves were not what the kernel waits for)
  sg.strip_w = (mode == 200 && g.tiles_x >= 2) ? 2 : 0;
  const bool zorder = mode == 500 && g.tiles_x == g.tiles_y && (g.tiles_x & (g.tiles_x - 1)) == 0 && g.tiles_x <= 256;  // (A/B)
  sg.full_strips = sg.strip_w ? g.tiles_x / sg.strip_w : 0;
  sg.rem_w = sg.strip_w ? g.tiles_x - sg.full_strips * sg.strip_w : 0;
  sg.inv_per_strip = sg.strip_w ? 1.f / (float)(sg.strip_w * g.tiles_y) : 0.f;
  sg.inv_strip_w = sg.strip_w ? 1.f / (float)sg.strip_w : 0.f;
  sg.inv_rem_w = sg.rem_w ? 1.f / (float)sg.rem_w : 0.f;
  const bool all_full = exact && g.W[2] % kQF == 0 && g.H[2] % kQF == 0;  // e.g. every input whose sides are multiples of 128
  const bool v_hm = (lanes & 2) != 0;  // value stored head-major, (heads, B, S, 32)
  sg.v_pix = v_hm ? 32 : heads * 32;
  sg.v_head = v_hm ? (long long)B * S * 32 : 32;
  sg.v_img = v_hm ? (long long)S * 32 : (long long)S * heads * 32;
  const bool ln = FUSED && (lanes & 1) != 0;
  auto kfn = ln ? msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 1> : msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 0>;
  if (all_full) kfn = ln ? msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 9> : msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 8>;
  if constexpr (FUSED) if (slab) {
    kfn = all_full ? msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 9 + 128> : msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 1 + 128>;
#ifdef WM2F_PROFILING
    if (mode == 801 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 9 + 128 + 256>;
    if (mode == 802 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 9 + 128 + 512>;
    if (mode == 803 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 9 + 128 + 256 + 512>;
    if (mode == 807 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 7, 0, 8, true, 9 + 128>;
    // timing ablations of the slab-order kernel (OUTPUTS NOT VALID): 814 no LDS reads, 815 no window DMA, 816 no operand loads / stores
    if (mode == 817 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 9 + 128 + 2>;      // slab + round-1 loader schedule
    if (mode == 820 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 8, 0, 8, true, 9 + 128>;  // ablation: operand loads from one hot record
    if (mode == 814 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 4, 0, 8, true, 9 + 128>;
    if (mode == 815 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 5, 0, 8, true, 9 + 128>;
    if (mode == 816 && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 6, 0, 8, true, 9 + 128>;
#endif
  }
  int threads = SCfg<8>::THREADS;
  if (!exact) kfn = ln ? msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, false, 1> : msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, false, 0>;
  // lanes bits 3 / 4: bf16 [offsets | logits] rows / bf16 output (the training forward under bf16 autocast: both or neither)
  if constexpr (FUSED) if ((lanes & 24) == 24) {
    if (ln || slab || mode != 0) return WM2F_OK;
    kfn = !exact ? msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, false, 1024 + 2048>
                 : (all_full ? msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 8 + 1024 + 2048> : msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 1024 + 2048>);
  } else if (lanes & 24) return WM2F_OK;
#ifndef WM2F_PROFILING
  if (mode != 0) return WM2F_OK;  // every other mode is a measured negative, an ablation or a stamped build: profiling library
#else
  if (half) {
    kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 4, true, 0>;
    threads = SCfg<4>::THREADS;
  }
  if (mode == 4 && exact) kfn = msdeform_stream_fwd_kernel<FUSED, 4, 0, 8, true, 0>;
  if (mode == 7 && exact) kfn = ln ? msdeform_stream_fwd_kernel<FUSED, 7, 0, 8, true, 1> : msdeform_stream_fwd_kernel<FUSED, 7, 0, 8, true, 0>;
  if (mode == 74 && exact) kfn = msdeform_stream_fwd_kernel<FUSED, 7, 0, 4, true, 0>;
  if (mode == 100 && exact) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 1, 8, true, 0>;  // flags instead of barriers
  if (mode == 200 && exact) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 4>;  // 2-wide vertical strips (A/B measurement)
  if (mode == 200 && exact && ln) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 5>;  // strips on the lane-major rows (in-model A/B)
  if (zorder && exact && ln && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 25>;  // Z-order tile walk (in-model A/B)
  if (mode == 600 && exact && ln && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 41>;  // younger gather waves at priority 1
  if (mode == 700 && exact && ln && all_full) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 73>;  // older gather waves at priority 1
  if (mode == 300 && exact && ln) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 3>;  // round-1 schedule on the lane-major rows
  else if (mode == 300 && exact) kfn = msdeform_stream_fwd_kernel<FUSED, 0, 0, 8, true, 2>;  // the round-1 loader schedule (A/B measurement)
#endif
  hipLaunchKernelGGL(kfn, dim3(wg), dim3(threads), 0, (hipStream_t)stream, (const float*)value, (const float*)a,
                     (const float*)b, (float*)out, sg, S, Q, heads);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: streaming launch failed: %s", who, hipGetErrorString(e));
    return WM2F_ELAUNCH;
  }
  *handled = true;
  return WM2F_OK;
}

template int launch_stream<false>(const void*, const void*, const void*, void*, const int32_t*, int, int, int, int, int,
                                  int, void*, const char*, bool*, int, int, int, int);
template int launch_stream<true>(const void*, const void*, const void*, void*, const int32_t*, int, int, int, int, int,
                                 int, void*, const char*, bool*, int, int, int, int);

}  // namespace wm2f

#ifdef WM2F_PROFILING
// Copy the MODE-7 time stamps (int64 [8192 workgroups][16 slots], s_memtime ticks) to host memory (include/wm2f_prof.h).
extern "C" int wm2f_debug_stamps(void* host_dst, int64_t n_bytes) {
  using namespace wm2f;
  WM2F_REQUIRE(host_dst && n_bytes > 0 && n_bytes <= (int64_t)sizeof(long long) * kStampGroups * kStampSlots,
               "wm2f_debug_stamps: bad destination / size");
  hipError_t e = hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), (size_t)n_bytes, 0, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    set_error("wm2f_debug_stamps: %s", hipGetErrorString(e));
    return WM2F_ELAUNCH;
  }
  return WM2F_OK;
}
#endif  // WM2F_PROFILING
