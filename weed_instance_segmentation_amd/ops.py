"""Python face of the libwm2f kernels: argument checking, pointer plumbing, autograd glue.

PyTorch is used for device memory, streams and autograd bookkeeping only.  Every function here
launches hand-written HIP through the C ABI (include/wm2f.h) on the caller's current stream and
RAISES if the tensors are not on a GPU or the library is missing -- there is no eager fallback.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch

from . import _lib
from ._lib import WM2F_BF16, WM2F_F32, check, host_i32, load


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self):
        self.records: dict[str, list] = {}

    def bracket(self, name, device):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.setdefault(name, []).append((a, b))
        return a, b

    def summary(self):
        """name -> (launches, mean microseconds); call after a device synchronize."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) * 1e3 / len(v)) for k, v in self.records.items()}


_timer: KernelTimer | None = None


def set_kernel_timer(t: KernelTimer | None) -> None:
    global _timer
    _timer = t


def _timed(name, tensor, fn):
    if _timer is None:
        return fn()
    a, b = _timer.bracket(name, tensor.device)
    a.record()
    r = fn()
    b.record()
    return r


def _p(t: torch.Tensor | None):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# Mixed precision (BASELINE configs 3-5 run under bf16 autocast): the fp32 entry points below get their inputs cast to
# fp32 with autocast switched off inside -- the policy PyTorch itself applies to grid_sample / softmax / layer_norm,
# which is what the dependency's K1 runs through.  The train step's own paths do not go through these casts: K1 takes
# the projection's rows in bf16 and writes bf16 (ms_deform_attn_rows), K2 and K3 run on the bf16 matrix cores
# (masked_xattn_bf16, mask_einsum_bf16), the token Linears' weight gradients read bf16 operands (token_wgrad).
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def _f32(t):
    """fp32 view of a half-precision tensor for the entry points without autograd (inference, matcher)."""
    return t.float() if isinstance(t, torch.Tensor) and t.dtype in (torch.bfloat16, torch.float16) else t


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.Wm2fError(f"{name} is on {t.device}: the wm2f kernels run on a GPU only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------- K1
class _MSDeformAttn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, value, loc, attn_w, level_hw):
        value, loc, attn_w = _req(value, "value"), _req(loc, "loc"), _req(attn_w, "attn_w")
        B, S, H, D = value.shape
        _, Q, _, L, P, _ = loc.shape
        if loc.shape != (B, Q, H, L, P, 2) or attn_w.shape != (B, Q, H, L, P):
            raise ValueError(f"msdeform: shapes disagree: value {tuple(value.shape)} loc {tuple(loc.shape)} "
                             f"attn_w {tuple(attn_w.shape)}")
        out = torch.empty(B, Q, H * D, device=value.device, dtype=value.dtype)
        lv = host_i32([x for hw in level_hw for x in hw])
        with torch.cuda.device(value.device):
            check(_timed("msdeform_fwd", value, lambda: load().wm2f_msdeform_fwd(
                _p(value), _p(loc), _p(attn_w), _p(out), lv, B, S, Q, H, D, L, P, WM2F_F32, _stream(value))),
                "wm2f_msdeform_fwd")
        ctx.save_for_backward(value, loc, attn_w)
        ctx.level_hw = tuple(tuple(int(x) for x in hw) for hw in level_hw)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        value, loc, attn_w = ctx.saved_tensors
        g_value, g_loc, g_w = ms_deform_attn_bwd(value, ctx.level_hw, loc, attn_w, grad_out)
        return g_value, g_loc, g_w, None


# K1 backward flavour: None = follow torch.are_deterministic_algorithms_enabled(); True / False force it.
K1_BWD_DETERMINISTIC: bool | None = None


def ms_deform_attn_bwd(value, level_hw, loc, attn_w, grad_out, deterministic: bool | None = None):
    """K1 backward (the autograd of HF:798-837): (grad_value, grad_loc, grad_attn_w).  `deterministic`: the fixed-point
    form of the grad_value scatter (wm2f_msdeform_bwd_det, run-to-run identical); default: module flag
    K1_BWD_DETERMINISTIC, else torch's deterministic-algorithms switch.  Shapes the fixed-point form does not cover
    raise under that switch (as torch's own ops without a deterministic form do) unless it is in warn-only mode."""
    value, loc, attn_w = _req(value, "value"), _req(loc, "loc"), _req(attn_w, "attn_w")
    grad_out = _req(grad_out, "grad_out")
    B, S, H, D = value.shape
    _, Q, _, L, P, _ = loc.shape
    if deterministic is None:
        deterministic = K1_BWD_DETERMINISTIC
    if deterministic is None:
        deterministic = torch.are_deterministic_algorithms_enabled()
    g_loc = torch.empty_like(loc)
    g_w = torch.empty_like(attn_w)
    lv = host_i32([x for hw in level_hw for x in hw])
    with torch.cuda.device(value.device):
        if deterministic:
            g_value = torch.empty_like(value)
            ws = torch.empty(max(16, int(load().wm2f_msdeform_bwd_det_workspace(lv, B, S, H, D, L))), device=value.device, dtype=torch.uint8)
            rc = _timed("msdeform_bwd_det", value, lambda: load().wm2f_msdeform_bwd_det(
                _p(value), _p(loc), _p(attn_w), _p(grad_out), _p(g_value), _p(g_loc), _p(g_w), _p(ws), lv, B, S, Q, H, D, L, P,
                WM2F_F32, _stream(value)))
            if rc != _lib.WM2F_EUNSUPPORTED:
                check(rc, "wm2f_msdeform_bwd_det")
                return g_value, g_loc, g_w
            if not torch.is_deterministic_algorithms_warn_only_enabled():
                raise RuntimeError("ms_deform_attn backward: no deterministic form for this shape (" + (load().wm2f_last_error() or b"?").decode() + ")")
        g_value = torch.zeros_like(value)
        check(_timed("msdeform_bwd", value, lambda: load().wm2f_msdeform_bwd(
            _p(value), _p(loc), _p(attn_w), _p(grad_out), _p(g_value), _p(g_loc), _p(g_w), lv, B, S, Q, H, D, L, P,
            WM2F_F32, _stream(value))), "wm2f_msdeform_bwd")
    return g_value, g_loc, g_w


def ms_deform_attn(value: torch.Tensor, level_hw: Sequence[Sequence[int]], loc: torch.Tensor,
                   attn_w: torch.Tensor) -> torch.Tensor:
    """K1 -- multi_scale_deformable_attention (HF:798-837).
    value (B,S,heads,D), loc (B,Q,heads,L,P,2), attn_w (B,Q,heads,L,P) -> (B,Q,heads*D)."""
    return _MSDeformAttn.apply(value, loc, attn_w, level_hw)


def k1_rows_applies(value: torch.Tensor, rows: torch.Tensor, level_hw, heads: int, n_points: int = 4) -> bool:
    """Host-side copy of the shape test of wm2f_msdeform_rows_fwd / _bwd (K1 for training on the merged projection's rows):
    the streaming kernel's shapes (3 levels 1 : 2 : 4 coarse first, 4 points, head_dim 32, queries == tokens), an even head
    count, rows fp32 or bf16, and no deterministic-algorithms request (the fixed-point grad_value form takes loc / attn_w)."""
    if not (value.is_cuda and rows.is_cuda and value.dim() == 4 and rows.dim() == 3) or heads % 2:
        return False
    B, S, H, D = value.shape
    if H != heads or rows.shape != (B, S, heads * 3 * n_points * 3) or rows.dtype not in (torch.float32, torch.bfloat16):
        return False
    if value.dtype not in (torch.float32, rows.dtype):
        return False
    det = K1_BWD_DETERMINISTIC if K1_BWD_DETERMINISTIC is not None else torch.are_deterministic_algorithms_enabled()
    return (not det) and k1_lanes_applies(level_hw, S, D, n_points, B, heads)


class _MSDeformAttnRows(torch.autograd.Function):
    """K1 with the prologue of HF:983-1002 inside, differentiable: (value, rows = [offsets | logits]) -> out, with the backward
    kernels writing the ROW gradient directly (wm2f_msdeform_rows_fwd / _bwd).  rows / out / their gradients share one dtype
    (fp32, or bf16 under bf16 autocast); value is cast to fp32 once (the kernels' windows are fp32) and kept for the backward."""

    @staticmethod
    def forward(ctx, value, rows, level_hw, heads):
        B, S, H, D = value.shape
        v32 = _req(value if value.dtype == torch.float32 else value.float(), "value")
        rows = _req(rows, "rows", rows.dtype)
        lp = rows.dtype == torch.bfloat16
        out = torch.empty(B, S, H * D, device=value.device, dtype=rows.dtype)
        lv = host_i32([x for hw in level_hw for x in hw])
        with torch.cuda.device(value.device):
            check(_timed("msdeform_rows_fwd", v32, lambda: load().wm2f_msdeform_rows_fwd(
                _p(v32), _p(rows), _p(out), lv, B, S, S, H, D, 3, 4, WM2F_BF16 if lp else WM2F_F32, _stream(v32))),
                "wm2f_msdeform_rows_fwd")
        ctx.save_for_backward(v32, rows)
        ctx.level_hw = tuple(tuple(int(x) for x in hw) for hw in level_hw)
        ctx.value_dtype = value.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        v32, rows = ctx.saved_tensors
        B, S, H, D = v32.shape
        lp = rows.dtype == torch.bfloat16
        grad_out = _req(grad_out if grad_out.dtype == rows.dtype else grad_out.to(rows.dtype), "grad_out", rows.dtype)
        g_value = torch.empty_like(v32)  # (cleared by the backward's first kernel)
        g_rows = torch.empty_like(rows)
        lv = host_i32([x for hw in ctx.level_hw for x in hw])
        with torch.cuda.device(v32.device):
            check(_timed("msdeform_rows_bwd", v32, lambda: load().wm2f_msdeform_rows_bwd(
                _p(v32), _p(rows), _p(grad_out), _p(g_value), _p(g_rows), lv, B, S, S, H, D, 3, 4,
                WM2F_BF16 if lp else WM2F_F32, _stream(v32))), "wm2f_msdeform_rows_bwd")
        return (g_value if ctx.value_dtype == torch.float32 else g_value.to(ctx.value_dtype)), g_rows, None, None


def ms_deform_attn_rows(value: torch.Tensor, level_hw, rows: torch.Tensor, heads: int) -> torch.Tensor:
    """K1 on the merged projection's rows, with autograd (training): value (B,S,heads,32) fp32 / bf16, rows (B,S,heads*36) =
    [offsets (heads,3,4,2) | logits (heads,12)] fp32 / bf16 -> (B,S,heads*32) in the rows' dtype.  Reference points are the
    tokens' pixel centres (HF:1127-1156 with valid ratios of 1).  Check k1_rows_applies first."""
    if not k1_rows_applies(value, rows, level_hw, heads):
        raise ValueError("ms_deform_attn_rows: shapes / dtypes outside wm2f_msdeform_rows_fwd (see k1_rows_applies)")
    return _MSDeformAttnRows.apply(value, rows, level_hw, heads)


def ms_deform_attn_fused(value: torch.Tensor, level_hw, offsets: torch.Tensor, logits: torch.Tensor,
                         ref: torch.Tensor) -> torch.Tensor:
    """K1 with the softmax / location prologue of HF:983-1002 fused (inference path, no autograd).
    offsets (B,Q,heads,L,P,2) raw, logits (B,Q,heads,L*P) raw, ref (Q,L,2)."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (value, offsets, logits)):
        raise RuntimeError("ms_deform_attn_fused has no backward; use ms_deform_attn when training")
    value, offsets, logits, ref = (_req(_f32(value), "value"), _req(_f32(offsets), "offsets"), _req(_f32(logits), "logits"),
                                   _req(_f32(ref), "ref"))
    B, S, H, D = value.shape
    _, Q, _, L, P, _ = offsets.shape
    if logits.shape != (B, Q, H, L * P) or ref.shape != (Q, L, 2):
        raise ValueError("ms_deform_attn_fused: shapes disagree")
    out = torch.empty(B, Q, H * D, device=value.device, dtype=value.dtype)
    lv = host_i32([x for hw in level_hw for x in hw])
    with torch.cuda.device(value.device):
        check(_timed("msdeform_fused_fwd", value, lambda: load().wm2f_msdeform_fused_fwd(
            _p(value), _p(offsets), _p(logits), _p(ref), _p(out), lv, B, S, Q, H, D, L, P, WM2F_F32, _stream(value))),
            "wm2f_msdeform_fused_fwd")
    return out


def ms_deform_attn_fused_packed(value, level_hw, packed, ref, heads: int, L: int, P: int, margin: int = 4):
    """Inference K1 fed by ONE merged projection: packed (B,Q,heads*L*P*3) = [offsets | logits] per token.
    Falls back to the two-array fused kernel (after splitting) where the LDS-window kernel does not apply."""
    if torch.is_grad_enabled() and (value.requires_grad or packed.requires_grad):
        raise RuntimeError("ms_deform_attn_fused_packed has no backward; use ms_deform_attn when training")
    value, packed, ref = _req(_f32(value), "value"), _req(_f32(packed), "packed"), _f32(ref)
    B, S, H, D = value.shape
    Q = packed.shape[1]
    if H != heads or packed.shape != (B, Q, heads * L * P * 3):
        raise ValueError(f"ms_deform_attn_fused_packed: value {tuple(value.shape)} packed {tuple(packed.shape)}")
    out = torch.empty(B, Q, H * D, device=value.device, dtype=value.dtype)
    lv = host_i32([x for hw in level_hw for x in hw])
    with torch.cuda.device(value.device):
        rc = _timed("msdeform_fused_fwd", value, lambda: load().wm2f_msdeform_fused_packed_fwd(
            _p(value), _p(packed), _p(out), lv, B, S, Q, H, D, L, P, WM2F_F32, int(margin), _stream(value)))
    if rc == _lib.WM2F_EUNSUPPORTED:  # shape outside the LDS-window kernels -> direct-gather HIP kernel
        n_off = heads * L * P * 2
        off = packed[..., :n_off].reshape(B, Q, heads, L, P, 2).contiguous()
        logits = packed[..., n_off:].reshape(B, Q, heads, L * P).contiguous()
        return ms_deform_attn_fused(value, level_hw, off, logits, ref)
    check(rc, "wm2f_msdeform_fused_packed_fwd")
    return out


def k1_lanes_applies(level_hw, n_tokens: int, head_dim: int, n_points: int, batch: int = 1, heads: int = 8) -> bool:
    """Host-side copy of the shape test of wm2f_msdeform_fused_lanes_fwd (the streaming kernel): 3 levels with sides
    exactly 1 : 2 : 4 coarse first, 4 points, head_dim 32, queries == tokens, 32-bit offsets."""
    if len(level_hw) != 3 or n_points != 4 or head_dim != 32:
        return False
    (h0, w0), (h1, w1), (h2, w2) = [(int(a), int(b)) for a, b in level_hw]
    if (h1, w1) != (2 * h0, 2 * w0) or (h2, w2) != (4 * h0, 4 * w0) or h0 < 1 or w0 < 1 or 21 * h0 * w0 != n_tokens:
        return False
    row = heads * 3 * 4 * 3 * 4
    return batch * n_tokens < (1 << 24) and batch * n_tokens * row < 0x7fffffff and batch * n_tokens * heads * 128 < 0x7fffffff


def k1_lane_order(heads: int) -> torch.Tensor:
    """Row permutation that turns the [sampling_offsets ; attention_weights] projection (heads*24 offset rows, then heads*12
    logit rows; L = 3, P = 4) into the kernel's record order (include/wm2f.h, wm2f_msdeform_fused_lanes_fwd): 36 numbers per
    head in 16-byte pieces -- [x0 y0 x1 y1] of lanes (= point slots) 0..3, [x2 y2 w0 w1] of lanes 0..3, w2 of lanes 0..3."""
    L, P = 3, 4
    n_off = heads * L * P * 2
    h = torch.arange(heads)[:, None]
    j = torch.arange(P)[None, :]
    off = lambda l, xy: ((h * L + l) * P + j) * 2 + xy       # (heads, P) row index of offsets[h, l, j, xy]
    lg = lambda l: n_off + h * (L * P) + l * P + j             # (heads, P) row index of logits[h, l * P + j]
    a = torch.stack([off(0, 0), off(0, 1), off(1, 0), off(1, 1)], -1).reshape(heads, 16)
    b = torch.stack([off(2, 0), off(2, 1), lg(0), lg(1)], -1).reshape(heads, 16)
    return torch.cat([a, b, lg(2)], 1).reshape(-1)


def k1_lane_rows(offsets: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """(B, S, heads, 3, 4, 2) offsets and (B, S, heads, 12) logits -> (B, S, heads * 36) rows in the kernel's record order
    (what the merged projection with its rows permuted by `k1_lane_order` writes); tests and tools build operands with it."""
    B, S, H = offsets.shape[:3]
    packed = torch.cat([offsets.reshape(B, S, -1), logits.reshape(B, S, -1)], -1)
    return packed[..., k1_lane_order(H).to(packed.device)].contiguous()


def ms_deform_attn_fused_lanes(value, level_hw, lanes, heads: int, head_major: bool = False, value_head_major: bool = False,
                               slab_order: bool = False):
    """Inference K1 fed by ONE merged projection whose rows are in lane-major order (include/wm2f.h,
    wm2f_msdeform_fused_lanes_fwd): lanes (B,Q,heads*36), or head-major (heads,B,Q,36) -- what token_linear(out_group=36)
    writes and the kernel reads in fewer cache lines.  Streaming kernel only -- check `k1_lanes_applies` first; a shape it
    does not take RAISES (no silent re-route: the caller owns the row order of its projection)."""
    if torch.is_grad_enabled() and (value.requires_grad or lanes.requires_grad):
        raise RuntimeError("ms_deform_attn_fused_lanes has no backward; use ms_deform_attn when training")
    value, lanes = _req(_f32(value), "value"), _req(_f32(lanes), "lanes")
    if value_head_major:  # (heads, B, S, D): what token_linear(out_group=D) writes
        H, B, S, D = value.shape
    else:
        B, S, H, D = value.shape
    Q = lanes.shape[2] if head_major else lanes.shape[1]
    if H != heads or tuple(lanes.shape) != ((heads, B, Q, 36) if head_major else (B, Q, heads * 36)):
        raise ValueError(f"ms_deform_attn_fused_lanes: value {tuple(value.shape)} lanes {tuple(lanes.shape)} head_major={head_major}")
    out = torch.empty(B, Q, H * D, device=value.device, dtype=value.dtype)
    lv = host_i32([x for hw in level_hw for x in hw])
    with torch.cuda.device(value.device):
        check(_timed("msdeform_fused_fwd", value, lambda: load().wm2f_msdeform_fused_lanes_fwd(
            _p(value), _p(lanes), _p(out), lv, B, S, Q, H, D, 3, 4, WM2F_F32, (1 if head_major else 0) | (2 if value_head_major else 0) | (4 if slab_order else 0),
            _stream(value))),
            "wm2f_msdeform_fused_lanes_fwd")
    return out


def ms_deform_attn_variant(value, level_hw, a, b, ref=None, fused=False, variant=0, margin=4) -> torch.Tensor:
    """K1 with the kernel variant exposed (no autograd): variant 0 auto, 1 direct gather, 2 LDS windows.
    fused=False: a = loc, b = attn_w.  fused=True: a = raw offsets, b = raw logits, ref (Q,L,2)."""
    value, a, b = _req(value, "value"), _req(a, "a"), _req(b, "b")
    if fused:
        ref = _req(ref, "ref")
    B, S, H, D = value.shape
    _, Q, _, L, P, _ = a.shape
    out = torch.empty(B, Q, H * D, device=value.device, dtype=value.dtype)
    lv = host_i32([x for hw in level_hw for x in hw])
    with torch.cuda.device(value.device):
        check(_timed(f"msdeform_v{variant}", value, lambda: load().wm2f_msdeform_fwd_v(
            _p(value), _p(a), _p(b), _p(ref if fused else None), _p(out), lv, B, S, Q, H, D, L, P, WM2F_F32,
            1 if fused else 0, int(variant), int(margin), _stream(value))), "wm2f_msdeform_fwd_v")
    return out


# ----------------------------------------------------------------------------------------- K3
class _MaskEinsum(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, emb, pix, tag=None):
        emb, pix = _req(emb, "emb"), _req(pix, "pix")
        B, Q, C = emb.shape
        if pix.dim() != 4 or pix.shape[0] != B or pix.shape[1] != C:
            raise ValueError(f"mask_einsum: emb {tuple(emb.shape)} vs pix {tuple(pix.shape)}")
        Hh, Ww = pix.shape[2:]
        out = torch.empty(B, Q, Hh, Ww, device=emb.device, dtype=emb.dtype)
        with torch.cuda.device(emb.device):
            check(_timed("mask_einsum_fwd" + (f"_{tag}" if tag else ""), emb, lambda: load().wm2f_mask_einsum_fwd(
                _p(emb), _p(pix), _p(out), B, Q, C, Hh * Ww, WM2F_F32, _stream(emb))), "wm2f_mask_einsum_fwd")
        ctx.save_for_backward(emb, pix)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        emb, pix = ctx.saved_tensors
        g_emb, g_pix = mask_einsum_bwd(emb, pix, grad_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return g_emb, g_pix, None


def mask_einsum_bwd_applies(Q: int, C: int, HW: int) -> bool:
    """Shapes the hand-written K3 backward covers (wm2f.h); others take two batched library GEMMs."""
    return C % 64 == 0 and Q % 4 == 0 and HW % 4 == 0 and (C + 16) * HW * 4 < 2 ** 31 and (Q + 16) * HW * 4 < 2 ** 31


def mask_einsum_bwd(emb, pix, grad_out, want_emb=True, want_pix=True):
    """K3 backward: g_emb (B,Q,C) = grad x pix^T, g_pix (B,C,H,W) = emb^T x grad (wm2f_mask_einsum_bwd; deterministic)."""
    B, Q, C = emb.shape
    HW = int(pix.shape[2]) * int(pix.shape[3])
    if not mask_einsum_bwd_applies(Q, C, HW):
        go = grad_out.reshape(B, Q, -1)
        g_emb = torch.bmm(go, pix.reshape(B, C, -1).transpose(1, 2)) if want_emb else None
        g_pix = torch.bmm(emb.transpose(1, 2), go).view_as(pix) if want_pix else None
        return g_emb, g_pix
    emb, pix, go = _req(emb, "emb"), _req(pix, "pix"), _req(grad_out.contiguous(), "grad_out")
    g_emb = torch.empty_like(emb) if want_emb else None
    g_pix = torch.empty_like(pix) if want_pix else None
    if not (want_emb or want_pix):
        return None, None
    with torch.cuda.device(emb.device):
        ws = torch.empty(int(load().wm2f_mask_einsum_bwd_workspace(B, Q, C, HW)), device=emb.device, dtype=torch.uint8)
        check(_timed("mask_einsum_bwd", emb, lambda: load().wm2f_mask_einsum_bwd(
            _p(emb), _p(pix), _p(go), _p(g_emb) if want_emb else None, _p(g_pix) if want_pix else None, _p(ws),
            B, Q, C, HW, WM2F_F32, _stream(emb))), "wm2f_mask_einsum_bwd")
    return g_emb, g_pix


def mask_einsum(emb: torch.Tensor, pix: torch.Tensor, tag: str | None = None) -> torch.Tensor:
    """K3 -- einsum('bqc,bchw->bqhw') (HF:2046) on the fp32 matrix cores.  `tag` only names the launch for the kernel timer."""
    return _MaskEinsum.apply(emb, pix, tag)


def nchw_to_pixel_major_bf16(pix: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) bf16 -> (B, H*W, C) bf16 (tiled transpose; no autograd: used as saved data of mask_einsum_bf16)."""
    pix = _req(pix.detach(), "pix", torch.bfloat16)
    B, C, Hh, Ww = pix.shape
    out = torch.empty(B, Hh * Ww, C, device=pix.device, dtype=torch.bfloat16)
    with torch.cuda.device(pix.device):
        check(load().wm2f_nchw_to_pixel_major_bf16(_p(pix), _p(out), B, C, Hh * Ww, _stream(pix)), "wm2f_nchw_to_pixel_major_bf16")
    return out


class _MaskEinsumBf16(torch.autograd.Function):
    """bf16 operands, fp32 logits.  `pix_t` is the pixel-major copy of `pix` (made once per forward)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, emb, pix, pix_t):
        emb = _req(emb if emb.dtype == torch.bfloat16 else emb.to(torch.bfloat16), "emb", torch.bfloat16)
        pix_t = _req(pix_t, "pix_t", torch.bfloat16)
        B, Q, C = emb.shape
        Hh, Ww = pix.shape[2:]
        if pix_t.shape != (B, Hh * Ww, C):
            raise ValueError(f"mask_einsum_bf16: pix_t {tuple(pix_t.shape)} vs emb {tuple(emb.shape)}, pix {tuple(pix.shape)}")
        out = torch.empty(B, Q, Hh, Ww, device=emb.device, dtype=torch.float32)
        with torch.cuda.device(emb.device):
            for q0 in range(0, Q, 112):  # the kernel holds at most 7 query tiles
                q1 = min(Q, q0 + 112)
                e = emb[:, q0:q1].contiguous() if (q0, q1) != (0, Q) else emb
                o = out[:, q0:q1] if (q0, q1) != (0, Q) else out
                oc = o if o.is_contiguous() else torch.empty(B, q1 - q0, Hh, Ww, device=emb.device, dtype=torch.float32)
                check(_timed("mask_einsum_bf16_fwd", emb, lambda: load().wm2f_mask_einsum_bf16_fwd(
                    _p(e), _p(pix_t), _p(oc), B, q1 - q0, C, Hh * Ww, _stream(emb))), "wm2f_mask_einsum_bf16_fwd")
                if oc is not o:
                    o.copy_(oc)
        # `pix` (the NCHW tensor) is the backward's right operand: contiguous along the pixel contraction.  It is an input
        # of the forward and alive anyway; pix_t is kept only for shapes the backward kernels do not cover.
        ctx.save_for_backward(emb, pix_t, pix)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        emb, pix_t, pix = ctx.saved_tensors
        g_emb, g_pix = mask_einsum_bf16_bwd(emb, pix, grad_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1], pix_t)
        return g_emb, g_pix, None


def mask_einsum_bf16_bwd_applies(Q: int, C: int, HW: int) -> bool:
    """Shapes the hand-written bf16 K3 backward covers (wm2f.h); others take two batched library GEMMs."""
    return C % 64 == 0 and Q % 4 == 0 and Q <= 112 and HW % 8 == 0 and (C + 16) * HW * 2 < 2 ** 31 and (Q + 16) * HW * 4 < 2 ** 31


def mask_einsum_bf16_bwd(emb, pix, grad_out, want_emb=True, want_pix=True, pix_t=None):
    """K3 backward under bf16 autocast: emb (B,Q,C) bf16, pix (B,C,H,W) bf16, grad_out (B,Q,H,W) fp32 -> g_emb, g_pix bf16
    (wm2f_mask_einsum_bf16_bwd: grad rounded to bf16 in registers, fp32 accumulation, deterministic)."""
    B, Q, C = emb.shape
    HW = int(pix.shape[2]) * int(pix.shape[3])
    if not (want_emb or want_pix):
        return None, None
    if not mask_einsum_bf16_bwd_applies(Q, C, HW) or pix.dtype != torch.bfloat16 or not pix.is_contiguous():
        go = grad_out.reshape(B, Q, -1).to(torch.bfloat16)
        if pix_t is None:
            pix_t = pix.reshape(B, C, HW).transpose(1, 2).to(torch.bfloat16)
        g_emb = torch.bmm(go, pix_t) if want_emb else None
        g_pix = torch.bmm(emb.transpose(1, 2), go).view(pix.shape).to(pix.dtype) if want_pix else None
        return g_emb, g_pix
    emb = _req(emb, "emb", torch.bfloat16)
    pix = _req(pix, "pix", torch.bfloat16)
    go = _req(grad_out.contiguous(), "grad_out")
    g_emb = torch.empty_like(emb) if want_emb else None
    g_pix = torch.empty_like(pix) if want_pix else None
    with torch.cuda.device(emb.device):
        ws = torch.empty(int(load().wm2f_mask_einsum_bf16_bwd_workspace(B, Q, C, HW)), device=emb.device, dtype=torch.uint8)
        check(_timed("mask_einsum_bf16_bwd", emb, lambda: load().wm2f_mask_einsum_bf16_bwd(
            _p(emb), _p(pix), _p(go), _p(g_emb) if want_emb else None, _p(g_pix) if want_pix else None, _p(ws),
            B, Q, C, HW, _stream(emb))), "wm2f_mask_einsum_bf16_bwd")
    return g_emb, g_pix


def mask_einsum_bf16(emb: torch.Tensor, pix: torch.Tensor, pix_t: torch.Tensor) -> torch.Tensor:
    """K3 under bf16 autocast: emb (B,Q,C), pix (B,C,H,W) bf16 (for shape and gradient), pix_t = nchw_to_pixel_major_bf16(pix)
    -> fp32 logits (B,Q,H,W)."""
    return _MaskEinsumBf16.apply(emb, pix, pix_t)


def mask_einsum_attn_mask(emb: torch.Tensor, pix_level: torch.Tensor, tag: str | None = None):
    """K3 with the thresholding epilogue fused (HF:2046, :2051-2053, :1912-1914): emb (B,Q,C), pix_level (B,C,h,w) = the
    mask features at the LEVEL's resolution -> (mask (B,Q,h*w) uint8 1 = blocked, row_open (B,Q) int32).  No logits are
    written; no autograd (the dependency detaches the mask, HF:2054)."""
    emb, pix_level = _req(_f32(emb.detach()), "emb"), _req(_f32(pix_level.detach()), "pix_level")
    B, Q, C = emb.shape
    if pix_level.dim() != 4 or pix_level.shape[0] != B or pix_level.shape[1] != C:
        raise ValueError(f"mask_einsum_attn_mask: emb {tuple(emb.shape)} vs pix {tuple(pix_level.shape)}")
    HW = int(pix_level.shape[2]) * int(pix_level.shape[3])
    mask = torch.empty(B, Q, HW, device=emb.device, dtype=torch.uint8)
    row_open = torch.empty(B, Q, device=emb.device, dtype=torch.int32)
    with torch.cuda.device(emb.device):
        check(_timed("mask_einsum_attn_mask" + (f"_{tag}" if tag else ""), emb, lambda: load().wm2f_mask_einsum_attn_mask_fwd(
            _p(emb), _p(pix_level), _p(mask), _p(row_open), B, Q, C, HW, WM2F_F32, _stream(emb))),
            "wm2f_mask_einsum_attn_mask_fwd")
    return mask, row_open


def attn_mask_build(logits: torch.Tensor, size: Sequence[int]):
    """HF:2048-2054 + HF:1912-1914: (mask (B,Q,Hn*Wn) uint8 1=blocked, row_open (B,Q) int32).  No grad."""
    logits = _req(_f32(logits.detach()), "logits")
    B, Q, H, W = logits.shape
    Hn, Wn = int(size[0]), int(size[1])
    mask = torch.empty(B, Q, Hn * Wn, device=logits.device, dtype=torch.uint8)
    row_open = torch.empty(B, Q, device=logits.device, dtype=torch.int32)
    with torch.cuda.device(logits.device):
        check(_timed("attn_mask_build", logits, lambda: load().wm2f_attn_mask_build(
            _p(logits), _p(mask), _p(row_open), B, Q, H, W, Hn, Wn, _stream(logits))), "wm2f_attn_mask_build")
    return mask, row_open


# ----------------------------------------------------------------------------------------- K2
class _MaskedXAttn(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, mask, row_open, heads):
        q, k, v = _req(q, "q"), _req(k, "k"), _req(v, "v")
        B, Q, E = q.shape
        N = k.shape[1]
        D = E // heads
        if k.shape != (B, N, E) or v.shape != (B, N, E) or D * heads != E:
            raise ValueError(f"masked_xattn: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} heads {heads}")
        if mask is not None:
            mask = _req(mask, "mask", torch.uint8)
            if mask.shape != (B, Q, N):
                raise ValueError(f"masked_xattn: mask {tuple(mask.shape)} != {(B, Q, N)}")
        if row_open is not None:
            row_open = _req(row_open, "row_open", torch.int32)
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, Q, device=q.device, dtype=torch.float32)
        lib = load()
        ws = torch.empty(int(lib.wm2f_masked_xattn_workspace(B, heads, Q, N, D)), device=q.device, dtype=torch.uint8)
        with torch.cuda.device(q.device):
            check(_timed(f"masked_xattn_fwd_N{N}", q, lambda: lib.wm2f_masked_xattn_fwd(
                _p(q), _p(k), _p(v), _p(mask), _p(row_open), _p(out), _p(lse), _p(ws), B, heads, Q, N, D, WM2F_F32,
                _stream(q))), "wm2f_masked_xattn_fwd")
        ctx.save_for_backward(q, k, v, mask, row_open, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        q, k, v, mask, row_open, out, lse = ctx.saved_tensors
        grad_out = _req(grad_out, "grad_out")
        B, Q, E = q.shape
        N, heads = k.shape[1], ctx.heads
        D = E // heads
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lib = load()
        ws = torch.empty(int(lib.wm2f_masked_xattn_bwd_workspace(B, heads, Q, N, D)), device=q.device, dtype=torch.uint8)
        with torch.cuda.device(q.device):
            check(_timed(f"masked_xattn_bwd_N{N}", q, lambda: lib.wm2f_masked_xattn_bwd(
                _p(q), _p(k), _p(v), _p(mask), _p(row_open), _p(out), _p(lse), _p(grad_out), _p(gq), _p(gk), _p(gv),
                _p(ws), B, heads, Q, N, D, WM2F_F32, _stream(q))), "wm2f_masked_xattn_bwd")
        return gq, gk, gv, None, None, None


def masked_xattn_bf16_applies(q, k, v, heads: int) -> bool:
    """Shapes wm2f_masked_xattn_bf16_fwd is built for: bf16 q / k / v on a GPU, head_dim 32, whole 16-key tiles."""
    if not (q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.dim() == 3 and k.dim() == 3):
        return False
    E, N = q.shape[-1], k.shape[1]
    return E == heads * 32 and N % 16 == 0 and N * E * 2 < (1 << 31) and q.shape[1] * N < (1 << 31)


class _MaskedXAttnBf16(torch.autograd.Function):
    """K2 on bf16 operands (what the in_proj Linears emit under bf16 autocast): forward on wm2f_masked_xattn_bf16_fwd, fp32
    output; backward on wm2f_masked_xattn_bf16_bwd (up to 112 queries; beyond that the fp32 kernel on fp32 copies of the
    saved operands); gradients return in bf16, the operands' dtype."""

    @staticmethod
    def forward(ctx, q, k, v, mask, row_open, heads):
        q, k, v = _req(q, "q", torch.bfloat16), _req(k, "k", torch.bfloat16), _req(v, "v", torch.bfloat16)
        B, Q, E = q.shape
        N = k.shape[1]
        D = E // heads
        if k.shape != (B, N, E) or v.shape != (B, N, E) or D * heads != E:
            raise ValueError(f"masked_xattn: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} heads {heads}")
        if mask is not None:
            mask = _req(mask, "mask", torch.uint8)
            if mask.shape != (B, Q, N):
                raise ValueError(f"masked_xattn: mask {tuple(mask.shape)} != {(B, Q, N)}")
        if row_open is not None:
            row_open = _req(row_open, "row_open", torch.int32)
        out = torch.empty(B, Q, E, device=q.device, dtype=torch.float32)
        lse = torch.empty(B, heads, Q, device=q.device, dtype=torch.float32)
        lib = load()
        ws = torch.empty(int(lib.wm2f_masked_xattn_workspace(B, heads, Q, N, D)), device=q.device, dtype=torch.uint8)
        with torch.cuda.device(q.device):
            check(_timed(f"masked_xattn_bf16_fwd_N{N}", q, lambda: lib.wm2f_masked_xattn_bf16_fwd(
                _p(q), _p(k), _p(v), _p(mask), _p(row_open), _p(out), _p(lse), _p(ws), B, heads, Q, N, D, _stream(q))),
                "wm2f_masked_xattn_bf16_fwd")
        ctx.save_for_backward(q, k, v, mask, row_open, out, lse)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, mask, row_open, out, lse = ctx.saved_tensors
        grad_out = _req(grad_out.float(), "grad_out")
        B, Q, E = q.shape
        N, heads = k.shape[1], ctx.heads
        D = E // heads
        lib = load()
        ws = torch.empty(int(lib.wm2f_masked_xattn_bwd_workspace(B, heads, Q, N, D)), device=q.device, dtype=torch.uint8)
        with torch.cuda.device(q.device):
            if Q <= 112:  # one query chunk: the bf16 backward kernel
                gq = torch.empty(B, Q, E, device=q.device, dtype=torch.float32)
                gk, gv = torch.empty_like(k), torch.empty_like(v)
                check(_timed(f"masked_xattn_bf16_bwd_N{N}", q, lambda: lib.wm2f_masked_xattn_bf16_bwd(
                    _p(q), _p(k), _p(v), _p(mask), _p(row_open), _p(out), _p(lse), _p(grad_out), _p(gq), _p(gk), _p(gv),
                    _p(ws), B, heads, Q, N, D, _stream(q))), "wm2f_masked_xattn_bf16_bwd")
                return gq.to(torch.bfloat16), gk, gv, None, None, None
            # more queries (config 4: 200): the fp32 kernel on fp32 copies of the same bf16 values
            q, k, v = q.float(), k.float(), v.float()
            gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            check(_timed(f"masked_xattn_bwd_N{N}", q, lambda: lib.wm2f_masked_xattn_bwd(
                _p(q), _p(k), _p(v), _p(mask), _p(row_open), _p(out), _p(lse), _p(grad_out), _p(gq), _p(gk), _p(gv),
                _p(ws), B, heads, Q, N, D, WM2F_F32, _stream(q))), "wm2f_masked_xattn_bwd")
        return gq.to(torch.bfloat16), gk.to(torch.bfloat16), gv.to(torch.bfloat16), None, None, None


def masked_xattn(q, k, v, mask, row_open, heads: int) -> torch.Tensor:
    """K2 -- softmax(bias + q k^T) v with the shared byte mask (HF:1644-1650, TORCHF:6578-6600).
    q (B,Q,E) pre-scaled by 1/sqrt(D); k, v (B,N,E); mask (B,Q,N) uint8 or None; row_open (B,Q) int32 or None.
    bf16 q / k / v (bf16 autocast) with head_dim 32 and whole 16-key tiles run on the bf16 matrix cores and return fp32;
    everything else computes in fp32."""
    if masked_xattn_bf16_applies(q, k, v, heads):
        with torch.autocast("cuda", enabled=False):
            return _MaskedXAttnBf16.apply(q, k, v, mask, row_open, heads)
    return _MaskedXAttn.apply(q, k, v, mask, row_open, heads)


# ----------------------------------------------------------------------------------------- K4
def matcher_cost(mask_logits, class_logits, tgt_masks, tgt_counts, tgt_classes, points, w_class, w_mask, w_dice):
    """K4 -- all cost matrices of a step in one go (HF:444-472), no sync.

    mask_logits (NL,B,Q,h,w) or a list of NL (B,Q,h,w) tensors (not stacked); class_logits (NL,B,Q,C1); tgt_masks (sum T, Ht, Wt) fp32 or uint8;
    tgt_counts: python list of T_i; tgt_classes (sum T,) int64; points (NL,B,P,2).
    Returns cost (NL,B,Q,Tmax) fp32 on the device; columns >= T_i are zero."""
    class_logits, points = _req(_f32(class_logits), "class_logits"), _req(_f32(points), "points")
    levels = None
    if isinstance(mask_logits, (list, tuple)):  # one (B,Q,h,w) tensor per level, used where it is
        levels = [_req(_f32(m), "mask level") for m in mask_logits]
        mask_logits = levels[0]
        NL, (B, Q, h, w) = len(levels), mask_logits.shape
        if any(m.shape != mask_logits.shape for m in levels):
            raise ValueError("matcher_cost: level tensors of different shapes")
    else:
        mask_logits = _req(_f32(mask_logits), "mask_logits")
        NL, B, Q, h, w = mask_logits.shape
    C1 = class_logits.shape[-1]
    P = points.shape[2]
    if tgt_masks.dtype == torch.bool:
        tgt_masks = tgt_masks.view(torch.uint8)
    tdt = 1 if tgt_masks.dtype == torch.uint8 else 0
    tgt_masks = _req(tgt_masks, "tgt_masks", torch.uint8 if tdt else torch.float32)
    tgt_classes = _req(tgt_classes, "tgt_classes", torch.int64)
    offs = [0]
    for t in tgt_counts:
        offs.append(offs[-1] + int(t))
    Tsum, Tmax = offs[-1], max([int(t) for t in tgt_counts] + [1])
    if tgt_masks.shape[0] != Tsum or tgt_classes.shape[0] != Tsum or len(tgt_counts) != B:
        raise ValueError("matcher_cost: target counts disagree with the target tensors")
    Ht, Wt = tgt_masks.shape[-2:]
    cost = torch.zeros(NL, B, Q, Tmax, device=mask_logits.device, dtype=torch.float32)
    lib = load()
    ws = torch.empty(max(int(lib.wm2f_matcher_workspace(NL, B, Q, P, Tsum)), 4), device=mask_logits.device, dtype=torch.uint8)
    with torch.cuda.device(mask_logits.device):
        if levels is not None:
            tab = (ctypes.c_void_p * NL)(*[m.data_ptr() for m in levels])
            check(_timed("matcher_cost", mask_logits, lambda: lib.wm2f_matcher_cost_levels(
                tab, _p(class_logits), _p(tgt_masks), tdt, host_i32(offs), _p(tgt_classes), _p(points), _p(cost), _p(ws), NL,
                B, Q, C1, h, w, Ht, Wt, P, Tmax, float(w_class), float(w_mask), float(w_dice), _stream(mask_logits))),
                "wm2f_matcher_cost_levels")
        else:
            check(_timed("matcher_cost", mask_logits, lambda: lib.wm2f_matcher_cost(
                _p(mask_logits), _p(class_logits), _p(tgt_masks), tdt, host_i32(offs), _p(tgt_classes), _p(points), _p(cost),
                _p(ws), NL, B, Q, C1, h, w, Ht, Wt, P, Tmax, float(w_class), float(w_mask), float(w_dice),
                _stream(mask_logits))), "wm2f_matcher_cost")
    return cost


# ----------------------------------------------------------------------------------------- point sampling
def lsa_batched(cost: torch.Tensor, counts: torch.Tensor, t_cap: int):
    """Linear sum assignment of every (level, image) cost matrix on the device (HF:474 without the host round trip).
    cost (NL, B, Q, Tmax) fp32; counts (B) int32 on the device = targets per image.  Returns rows, cols (NL, B, t_cap) int32:
    the min(Q, T_b) matched (query, target) pairs per problem sorted by query, bit-identical to scipy's
    linear_sum_assignment(cost[l, b, :, :T_b]); entries beyond min(Q, T_b) are unspecified."""
    cost = _req(cost, "cost")
    counts = _req(counts, "counts", torch.int32)
    NL, B, Q, Tmax = cost.shape
    rows = torch.empty(NL, B, t_cap, device=cost.device, dtype=torch.int32)
    cols = torch.empty_like(rows)
    with torch.cuda.device(cost.device):
        check(_timed("lsa_batched", cost, lambda: load().wm2f_lsa_batched(_p(cost), _p(counts), _p(rows), _p(cols), NL * B, B, Q, Tmax,
                                                                        int(t_cap), _stream(cost))), "wm2f_lsa_batched")
    return rows, cols


class _PointSample(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, feat, pts, map_index):
        tdt = 1 if feat.dtype in (torch.uint8, torch.bool) else 0
        if feat.dtype == torch.bool:
            feat = feat.view(torch.uint8)
        feat = _req(feat, "feat", torch.uint8 if tdt else torch.float32)
        pts = _req(pts, "pts")
        N, H, W = feat.shape
        M, P = pts.shape[:2]
        if pts.dim() != 3 or pts.shape[2] != 2:
            raise ValueError(f"point_sample: pts {tuple(pts.shape)}")
        if map_index is None:
            if M != N:
                raise ValueError(f"point_sample: {M} point rows for {N} maps and no map_index")
        else:
            map_index = _req(map_index, "map_index", torch.int32)
            if map_index.shape != (M,):
                raise ValueError(f"point_sample: map_index {tuple(map_index.shape)} != ({M},)")
        out = torch.empty(M, P, device=feat.device, dtype=torch.float32)
        with torch.cuda.device(feat.device):
            check(load().wm2f_point_sample_fwd(_p(feat), tdt, _p(pts), _p(map_index), _p(out), M, H, W, P,
                                               _stream(feat)), "wm2f_point_sample_fwd")
        ctx.save_for_backward(pts, map_index)
        ctx.shape = (N, H, W)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        pts, map_index = ctx.saved_tensors
        N, H, W = ctx.shape
        grad_out = _req(grad_out, "grad_out")
        g = torch.zeros(N, H, W, device=pts.device, dtype=torch.float32)
        with torch.cuda.device(pts.device):
            check(load().wm2f_point_sample_bwd(_p(grad_out), _p(pts), _p(map_index), _p(g), pts.shape[0], H, W,
                                               pts.shape[1], _stream(pts)), "wm2f_point_sample_bwd")
        return g, None, None


def point_sample(feat: torch.Tensor, pts: torch.Tensor, map_index: torch.Tensor | None = None) -> torch.Tensor:
    """sample_point (HF:245-274) for single-channel maps: feat (N,H,W), pts (M,P,2) in [0,1] (x,y) -> (M,P).
    Row m samples feat[map_index[m]] (int32) -- or feat[m] when map_index is None."""
    return _PointSample.apply(feat, pts, map_index)


# ----------------------------------------------------------------------------------------- fused passes
def bias_act_(x: torch.Tensor, bias: torch.Tensor, residual: torch.Tensor | None = None, relu: bool = True):
    """In place: x <- act(x + bias[c] (+ residual)) for an NCHW tensor (inference, no autograd)."""
    if not x.is_contiguous():
        raise ValueError("bias_act_: x must be NCHW-contiguous")
    _req(x, "x"), _req(bias, "bias")
    N, C, H, W = x.shape
    if residual is not None:
        residual = _req(residual, "residual")
        if residual.shape != x.shape:
            raise ValueError("bias_act_: residual shape")
    with torch.cuda.device(x.device):
        check(load().wm2f_bias_act(_p(x), _p(bias), _p(residual), _p(x), N, C, H * W, 1 if relu else 0, _stream(x)),
              "wm2f_bias_act")
    return x


def add_broadcast(a: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """a (B, ...) + p (1, ...) -- the same trailing shape, broadcast over the batch (a level's positional embedding added to its
    tokens, HF `with_pos_embed`).  Inference only (no autograd); fp32, element count of a row a multiple of 4."""
    a, p = _req(a, "a"), _req(p, "p")
    if p.shape[0] != 1 or p.shape[1:] != a.shape[1:] or (a.numel() // a.shape[0]) % 4:
        raise ValueError(f"add_broadcast: a {tuple(a.shape)} p {tuple(p.shape)}")
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        check(load().wm2f_add_broadcast(_p(a), _p(p), _p(out), int(a.shape[0]), int(a.numel() // a.shape[0]), _stream(a)),
              "wm2f_add_broadcast")
    return out


def add_layernorm(x, residual, gamma, beta, eps: float, pos: torch.Tensor | None = None):
    """LayerNorm(x + residual) over the last dim (= 256); with `pos` (rows_per_image, 256) also returns
    out + pos broadcast over the batch.  Inference only (no autograd)."""
    x, gamma, beta = _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    C = x.shape[-1]
    rows = x.numel() // C
    if residual is not None:
        residual = _req(residual, "residual")
    out = torch.empty_like(x)
    out_pos, pos_rows = None, 0
    if pos is not None:
        pos = _req(pos, "pos")
        pos_rows = pos.numel() // C
        if rows % pos_rows:
            raise ValueError("add_layernorm: pos rows do not divide x rows")
        out_pos = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(load().wm2f_add_layernorm(_p(x), _p(residual), _p(gamma), _p(beta), _p(pos), _p(out), _p(out_pos), rows, C,
                                        pos_rows, float(eps), _stream(x)), "wm2f_add_layernorm")
    return (out, out_pos) if pos is not None else out


class _AddLayerNormTrain(torch.autograd.Function):
    """LayerNorm(x + res) with its consumers' tensors from the same pass, and ONE backward pass (include/wm2f.h,
    wm2f_add_layernorm_train_*).  Returns (y fp32, y in bf16 or None, y + pos in bf16 / fp32 or None)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, pos, want_lp, yp_bf16, clamp):
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"add_layernorm_train: x is {x.dtype}")
        x = _req(x, "x", x.dtype)
        gamma, beta = _req(gamma, "gamma"), _req(beta, "beta")
        C = x.shape[-1]
        rows = x.numel() // C
        if res is not None:
            res = _req(res, "residual")
            if res.shape != x.shape:
                raise ValueError("add_layernorm_train: residual shape")
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        y_lp = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if want_lp else None
        yp, pos_rows = None, 0
        if pos is not None:
            pos = _req(pos, "pos")
            pos_rows = pos.numel() // C
            if rows % pos_rows:
                raise ValueError("add_layernorm_train: pos rows do not divide the token count")
            yp = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16 if yp_bf16 else torch.float32)
        stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
        xd = _lib.WM2F_BF16 if x.dtype == torch.bfloat16 else WM2F_F32
        with torch.cuda.device(x.device):
            check(_timed("add_layernorm_train_fwd", x, lambda: load().wm2f_add_layernorm_train_fwd(
                _p(x), xd, _p(res), _p(gamma), _p(beta), _p(pos), _p(y), _p(y_lp), _p(yp), _lib.WM2F_BF16 if yp_bf16 else WM2F_F32,
                _p(stats), rows, C, pos_rows, float(eps), float(clamp), _stream(x))), "wm2f_add_layernorm_train_fwd")
        ctx.save_for_backward(x, res, gamma, stats)
        ctx.pos_shape = None if pos is None else tuple(pos.shape)
        ctx.pos_needs_grad = pos is not None and pos.requires_grad
        outs = (y,) + ((y_lp,) if want_lp else ()) + ((yp,) if yp is not None else ())
        ctx.layout = (want_lp, yp is not None)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        x, res, gamma, stats = ctx.saved_tensors
        want_lp, has_yp = ctx.layout
        gy = grads[0]
        gy_lp = grads[1] if want_lp else None
        gyp = grads[1 + int(want_lp)] if has_yp else None
        C = x.shape[-1]
        rows = x.numel() // C
        if gy is None and gy_lp is None and gyp is None:
            return (None,) * 9
        gy = None if gy is None else _req(gy, "grad_y")
        gy_lp = None if gy_lp is None else _req(gy_lp, "grad_y_bf16", torch.bfloat16)
        if gyp is not None:
            gyp = _req(gyp, "grad_y_plus_pos", gyp.dtype)
        dsum = torch.empty(x.shape, device=x.device, dtype=torch.float32)  # the residual stream's gradient; x's too when x is fp32
        dx = torch.empty_like(x) if x.dtype == torch.bfloat16 else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        xd = _lib.WM2F_BF16 if x.dtype == torch.bfloat16 else WM2F_F32
        with torch.cuda.device(x.device):
            ws = torch.empty(max(16, int(load().wm2f_add_layernorm_train_workspace(rows))), device=x.device, dtype=torch.uint8)
            check(_timed("add_layernorm_train_bwd", x, lambda: load().wm2f_add_layernorm_train_bwd(
                _p(x), xd, _p(res), _p(gamma), _p(stats), _p(gy), _p(gy_lp), _p(gyp),
                _lib.WM2F_BF16 if (gyp is not None and gyp.dtype == torch.bfloat16) else WM2F_F32, _p(dsum), _p(dx), _p(dgamma), _p(dbeta),
                _p(ws), rows, C, _stream(x))), "wm2f_add_layernorm_train_bwd")
        gpos = None
        if ctx.pos_needs_grad and gyp is not None:  # pos is shared by the batch: its gradient is the sum over the images
            gpos = torch.sum(gyp.reshape(-1, *ctx.pos_shape), 0, dtype=torch.float32)
        return (dx if dx is not None else dsum), (dsum if res is not None else None), dgamma, dbeta, None, gpos, None, None, None


def add_layernorm_train_applies(x: torch.Tensor, residual: torch.Tensor | None) -> bool:
    return (x.is_cuda and x.shape[-1] == 256 and x.dtype in (torch.float32, torch.bfloat16)
            and (residual is None or (residual.dtype == torch.float32 and residual.shape == x.shape)))


def add_layernorm_train(x, residual, gamma, beta, eps: float, pos: torch.Tensor | None = None, want_bf16: bool = False,
                        pos_bf16: bool = False, clamp: float = 0.0):
    """Training form of `add_layernorm`: y = LayerNorm(x + residual) (fp32), differentiable, with -- from the same pass -- y in
    bf16 (`want_bf16`: the next Linear's operand under bf16 autocast) and y + pos (`pos` (rows_per_image, 256): the next
    layer's `hidden + pos`, in bf16 with `pos_bf16`).  clamp > 0 limits y to [-clamp, clamp] (NaN untouched; its gradient
    passes through).  Returns (y, y_bf16 or None, y_plus_pos or None)."""
    with torch.autocast("cuda", enabled=False):
        outs = _AddLayerNormTrain.apply(x, residual, gamma, beta, float(eps), pos, bool(want_bf16), bool(pos_bf16), float(clamp))
    y = outs[0]
    y_lp = outs[1] if want_bf16 else None
    yp = outs[1 + int(want_bf16)] if pos is not None else None
    return y, y_lp, yp


def token_linear_applies(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes wm2f_token_linear_fwd is built for: fp32 on a GPU, N in {256, 288}, K a multiple of 64, x / out below 2 GiB."""
    N, K = weight.shape
    M = x.numel() // max(K, 1)
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and N in (256, 288) and K % 64 == 0
            and x.shape[-1] == K and M * K * 4 < (1 << 31) and M * N * 4 < (1 << 31))


def token_linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool = False, residual: torch.Tensor | None = None,
                 ln: tuple | None = None, pos: torch.Tensor | None = None, out_group: int = 0):
    """Linear over tokens with its epilogue fused (inference, no autograd): x (..., K) @ weight (N, K)^T + bias, then
    optional ReLU, optional LayerNorm(value + residual) with ln = (gamma, beta, eps), and with `pos` (rows_per_image, N)
    additionally out + pos broadcast over the batch.  Returns out, or (out, out + pos).
    out_group = G > 0: the result comes back feature-group major, (N // G, *x.shape[:-1], G) -- with G = 36 K1's operand rows
    head-major (ms_deform_attn_fused_lanes(..., head_major=True))."""
    x, weight, bias = _req(x, "x"), _req(weight, "weight"), _req(bias, "bias")
    N, K = weight.shape
    if x.shape[-1] != K or bias.shape != (N,):
        raise ValueError(f"token_linear: x {tuple(x.shape)} weight {tuple(weight.shape)} bias {tuple(bias.shape)}")
    M = x.numel() // K
    if out_group and (out_group % 4 or N % out_group or ln is not None):
        raise ValueError("token_linear: out_group must divide N, be a multiple of 4 and exclude the LayerNorm epilogue")
    out = (torch.empty(N // out_group, *x.shape[:-1], out_group, device=x.device, dtype=torch.float32) if out_group
           else torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32))
    gamma = beta = None
    eps = 0.0
    if ln is not None:
        gamma, beta, eps = _req(ln[0], "gamma"), _req(ln[1], "beta"), float(ln[2])
    if residual is not None:
        residual = _req(residual, "residual")
        if residual.numel() != M * N:
            raise ValueError("token_linear: residual shape")
    out_pos, pos_rows = None, 0
    if pos is not None:
        pos = _req(pos, "pos")
        pos_rows = pos.numel() // N
        if M % pos_rows:
            raise ValueError("token_linear: pos rows do not divide the token count")
        out_pos = torch.empty_like(out)
    with torch.cuda.device(x.device):
        check(_timed(f"token_linear_K{K}_N{N}" + ("_ln" if ln is not None else ""), x, lambda: load().wm2f_token_linear_fwd(
            _p(x), _p(weight), _p(bias), _p(residual), _p(gamma), _p(beta), _p(pos), _p(out), _p(out_pos), M, K, N, 1 if relu else 0,
            pos_rows, eps, int(out_group), _stream(x))), "wm2f_token_linear_fwd")
    return (out, out_pos) if pos is not None else out


def token_wgrad_applies(dy: torch.Tensor, x: torch.Tensor) -> bool:
    """Shapes wm2f_token_wgrad_bf16 / _f32 are built for: both operands bf16 or both fp32 on a GPU, feature counts multiples
    of 8, operands below 2 GiB."""
    N, K = dy.shape[-1], x.shape[-1]
    M = x.numel() // max(K, 1)
    es = x.element_size()
    return (dy.is_cuda and x.is_cuda and dy.dtype == x.dtype and x.dtype in (torch.bfloat16, torch.float32) and N % 8 == 0
            and K % 8 == 0 and dy.numel() == M * N and M * N * es < 0x7fffffff and M * K * es < 0x7fffffff)


def token_wgrad(dy: torch.Tensor, x: torch.Tensor, want_bias: bool = True):
    """Weight (and bias) gradient of a Linear over tokens: dy (..., N), x (..., K), both bf16 or both fp32 -> dw (N, K) fp32 =
    dy^T x and db (N) fp32 = column sums of dy (None without `want_bias`).  fp32 accumulation, deterministic (include/wm2f.h)."""
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"token_wgrad: {x.dtype}")
    dy, x = _req(dy, "dy", x.dtype), _req(x, "x", x.dtype)
    N, K = dy.shape[-1], x.shape[-1]
    M = x.numel() // K
    if dy.numel() != M * N:
        raise ValueError(f"token_wgrad: dy {tuple(dy.shape)} x {tuple(x.shape)}")
    dw = torch.empty(N, K, device=x.device, dtype=torch.float32)
    db = torch.empty(N, device=x.device, dtype=torch.float32) if want_bias else None
    bf = x.dtype == torch.bfloat16
    with torch.cuda.device(x.device):
        ws = torch.empty(max(16, int(load().wm2f_token_wgrad_workspace(M, N, K))), device=x.device, dtype=torch.uint8)
        fn = load().wm2f_token_wgrad_bf16 if bf else load().wm2f_token_wgrad_f32
        check(_timed(f"token_wgrad_{'bf16' if bf else 'f32'}_N{N}_K{K}", x, lambda: fn(
            _p(dy), _p(x), _p(dw), _p(db), _p(ws), M, N, K, _stream(x))), "wm2f_token_wgrad")
    return dw, db


class _TokenLinear(torch.autograd.Function):
    """nn.Linear over tokens with the weight gradient on wm2f_token_wgrad_*: forward and input gradient are the library's GEMMs
    (in bf16 under bf16 autocast, as autocast runs F.linear; in fp32 otherwise), dW / db come back in fp32 -- the parameters'
    dtype -- from ONE pass over dy and x."""

    @staticmethod
    def forward(ctx, x, weight, bias, bf16):
        cdt = torch.bfloat16 if bf16 else torch.float32
        xc, wc = x.to(cdt), weight.to(cdt)
        ctx.save_for_backward(xc, wc)
        ctx.x_dtype, ctx.has_bias, ctx.cdt = x.dtype, bias is not None, cdt
        return torch.nn.functional.linear(xc, wc, None if bias is None else bias.to(cdt))

    @staticmethod
    def backward(ctx, grad_out):
        xc, wc = ctx.saved_tensors
        g = grad_out.to(ctx.cdt).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(g, wc).to(ctx.x_dtype)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = token_wgrad(g, xc.contiguous(), want_bias=ctx.has_bias)
        return gx, gw, gb, None


def linear_tokens(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """F.linear for the token matrices of the pixel decoder's encoder layers.  In training on a GPU (fp32, or under bf16
    autocast) the weight-gradient product -- a 256 x 256 output with a contraction over every token of the batch, which a
    library GEMM runs on 16 of 256 CUs -- goes to wm2f_token_wgrad_*.  Everything else is plain F.linear."""
    amp = torch.is_autocast_enabled("cuda")
    bf16 = amp and torch.get_autocast_dtype("cuda") == torch.bfloat16
    if (torch.is_grad_enabled() and x.is_cuda and weight.requires_grad and weight.dtype == torch.float32
            and (bf16 or (not amp and x.dtype == torch.float32))
            and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0
            and x.numel() // weight.shape[1] * max(weight.shape) * (2 if bf16 else 4) < 0x7fffffff):
        with torch.autocast("cuda", enabled=False):
            return _TokenLinear.apply(x, weight, bias, bf16)
    return torch.nn.functional.linear(x, weight, bias)


def tokens_to_nchw(tokens: torch.Tensor, start: int, h: int, w: int) -> torch.Tensor:
    """tokens (B, S, C) rows [start, start + h*w) -> (B, C, h, w), tiled transpose (inference only, no autograd)."""
    tokens = _req(tokens, "tokens")
    B, S, C = tokens.shape
    out = torch.empty(B, C, h, w, device=tokens.device, dtype=torch.float32)
    with torch.cuda.device(tokens.device):
        check(load().wm2f_tokens_to_nchw(_p(tokens), _p(out), B, S, C, int(start), h * w, _stream(tokens)), "wm2f_tokens_to_nchw")
    return out


def group_norm_tokens_(x: torch.Tensor, bias: torch.Tensor | None, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                       tokens: torch.Tensor, start: int) -> torch.Tensor:
    """tokens[:, start:start+H*W, :] <- GroupNorm(x + bias) in token layout, for x (B, C, H, W) and tokens (B, S, C)
    (inference, no autograd): one level's input projection of the pixel decoder, HF:1341-1357."""
    if not x.is_contiguous() or not tokens.is_contiguous():
        raise ValueError("group_norm_tokens_: x and tokens must be contiguous")
    _req(x, "x"), _req(tokens, "tokens")
    gamma, beta = _req(gamma, "gamma"), _req(beta, "beta")
    if bias is not None:
        bias = _req(bias, "bias")
    B, C, H, W = x.shape
    if tokens.dim() != 3 or tokens.shape[0] != B or tokens.shape[2] != C:
        raise ValueError("group_norm_tokens_: tokens must be (B, S, C)")
    ws = torch.empty(2 * B * groups, device=x.device, dtype=torch.float64)
    with torch.cuda.device(x.device):
        check(load().wm2f_group_norm_tokens(_p(x), _p(bias), _p(gamma), _p(beta), _p(tokens), _p(ws), B, C, int(groups), H * W,
                                            int(tokens.shape[1]), int(start), float(eps), _stream(x)), "wm2f_group_norm_tokens")
    return tokens


def resize_bilinear(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) for an NCHW fp32 map (inference, no autograd)."""
    if not x.is_contiguous():
        raise ValueError("resize_bilinear: x must be NCHW-contiguous")
    _req(x, "x")
    N, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty(N, C, Ho, Wo, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(load().wm2f_resize_bilinear(_p(x), _p(y), N * C, H, W, Ho, Wo, _stream(x)), "wm2f_resize_bilinear")
    return y


def resize_pyramid(x: torch.Tensor):
    """(half, quarter, eighth)-size bilinear resizes of an NCHW fp32 map in ONE pass (H, W divisible by 8): bit for bit
    `resize_bilinear(x, (H/2, W/2))`, `(H/4, W/4)`, `(H/8, W/8)`.  Inference, no autograd."""
    if not x.is_contiguous():
        raise ValueError("resize_pyramid: x must be NCHW-contiguous")
    _req(x, "x")
    N, C, H, W = x.shape
    if H % 8 or W % 8:
        raise ValueError("resize_pyramid: H and W must be divisible by 8")
    ys = [torch.empty(N, C, H >> k, W >> k, device=x.device, dtype=torch.float32) for k in (1, 2, 3)]
    with torch.cuda.device(x.device):
        check(_timed("resize_pyramid", x, lambda: load().wm2f_resize_pyramid(_p(x), _p(ys[0]), _p(ys[1]), _p(ys[2]), N * C, H, W, _stream(x))),
              "wm2f_resize_pyramid")
    return ys


def bias_relu_maxpool(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """MaxPool2d(3, 2, 1)(ReLU(x + bias[c])) of an NCHW map in one pass (inference, no autograd): the ResNet stem tail."""
    if not x.is_contiguous():
        raise ValueError("bias_relu_maxpool: x must be NCHW-contiguous")
    _req(x, "x"), _req(bias, "bias")
    N, C, H, W = x.shape
    y = torch.empty(N, C, H // 2, W // 2, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(load().wm2f_bias_relu_maxpool(_p(x), _p(bias), _p(y), N, C, H, W, _stream(x)), "wm2f_bias_relu_maxpool")
    return y


def group_norm_act_(x: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                    up: torch.Tensor | None = None, relu: bool = False) -> torch.Tensor:
    """In place: x <- act(GroupNorm(x) (+ bilinear upsample of `up` to x's size, align_corners=False)) for an NCHW map
    (inference, no autograd) -- the GroupNorm tails of the FPN step, HF:1395-1405."""
    if not x.is_contiguous():
        raise ValueError("group_norm_act_: x must be NCHW-contiguous")
    _req(x, "x")
    gamma, beta = _req(gamma, "gamma"), _req(beta, "beta")
    B, C, H, W = x.shape
    Hs = Ws = 0
    if up is not None:
        up = _req(up, "up")
        if up.dim() != 4 or up.shape[:2] != x.shape[:2]:
            raise ValueError("group_norm_act_: up must be (B, C, Hs, Ws)")
        Hs, Ws = int(up.shape[2]), int(up.shape[3])
    ws = torch.empty(2 * B * groups, device=x.device, dtype=torch.float64)
    with torch.cuda.device(x.device):
        check(load().wm2f_group_norm_act(_p(x), _p(gamma), _p(beta), _p(up), _p(x), _p(ws), B, C, int(groups), H, W, Hs, Ws,
                                         float(eps), 1 if relu else 0, _stream(x)), "wm2f_group_norm_act")
    return x


# ------------------------------------------------------------------ instance post-processing
_GRID = (384, 384)  # the dependency's hard-coded intermediate size (image_processing_mask2former.py:680-682)


def instance_scores(mask_logits: torch.Tensor, qidx: torch.Tensor):
    """(sum of sigmoid over set pixels, number of set pixels) of each selected query's mask on the 384 x 384 grid."""
    mask_logits, qidx = _req(mask_logits, "mask_logits"), _req(qidx, "qidx", torch.int32)
    B, Q, h, w = mask_logits.shape
    K = qidx.shape[1]
    s = torch.empty(B, K, device=mask_logits.device, dtype=torch.float32)
    c = torch.empty_like(s)
    with torch.cuda.device(mask_logits.device):
        check(load().wm2f_instance_scores(_p(mask_logits), _p(qidx), _p(s), _p(c), B, Q, K, h, w, _GRID[0], _GRID[1],
                                          _stream(mask_logits)), "wm2f_instance_scores")
    return s, c


def instance_any(mask_logits, qidx, cand, size):
    mask_logits, qidx, cand = _req(mask_logits, "mask_logits"), _req(qidx, "qidx", torch.int32), _req(cand, "cand", torch.uint8)
    B, Q, h, w = mask_logits.shape
    K = qidx.shape[1]
    out = torch.empty(B, K, device=mask_logits.device, dtype=torch.int32)
    with torch.cuda.device(mask_logits.device):
        check(load().wm2f_instance_any(_p(mask_logits), _p(qidx), _p(cand), _p(out), B, Q, K, h, w, _GRID[0], _GRID[1],
                                       int(size[0]), int(size[1]), _stream(mask_logits)), "wm2f_instance_any")
    return out


def instance_segmentation(mask_logits, kept_q, n_kept, size):
    mask_logits = _req(mask_logits, "mask_logits")
    kept_q, n_kept = _req(kept_q, "kept_q", torch.int32), _req(n_kept, "n_kept", torch.int32)
    B, Q, h, w = mask_logits.shape
    K = kept_q.shape[1]
    seg = torch.empty(B, int(size[0]), int(size[1]), device=mask_logits.device, dtype=torch.float32)
    with torch.cuda.device(mask_logits.device):
        check(load().wm2f_instance_segmentation(_p(mask_logits), _p(kept_q), _p(n_kept), _p(seg), B, Q, K, h, w, _GRID[0],
                                                _GRID[1], int(size[0]), int(size[1]), _stream(mask_logits)),
              "wm2f_instance_segmentation")
    return seg


def instance_maps(image_logits, kept_q, n, size):
    image_logits, kept_q = _req(image_logits, "image_logits"), _req(kept_q, "kept_q", torch.int32)
    Q, h, w = image_logits.shape
    maps = torch.empty(n, int(size[0]), int(size[1]), device=image_logits.device, dtype=torch.float32)
    with torch.cuda.device(image_logits.device):
        check(load().wm2f_instance_maps(_p(image_logits), _p(kept_q), int(n), _p(maps), h, w, _GRID[0], _GRID[1],
                                        int(size[0]), int(size[1]), _stream(image_logits)), "wm2f_instance_maps")
    return maps


# ------------------------------------------------------ point-sampled mask loss over all levels
def _ptr_table(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _PointSampleLevels(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, pts, index, neg_abs, unique, *maps):
        maps = [_req(m, "level map") for m in maps]
        pts, index = _req(pts, "pts"), _req(index, "index", torch.int32)
        NL, M, P = pts.shape[:3]
        N, H, W = maps[0].shape
        if len(maps) != NL or index.shape != (NL, M) or any(m.shape != (N, H, W) for m in maps):
            raise ValueError("point_sample_levels: one (N,H,W) map per level, pts (NL,M,P,2), index (NL,M)")
        out = torch.empty(NL, M, P, device=pts.device, dtype=torch.float32)
        with torch.cuda.device(pts.device):
            check(_timed("point_sample_levels_fwd", pts, lambda: load().wm2f_point_sample_levels_fwd(
                _ptr_table(maps), NL, _p(pts), _p(index), _p(out), M, H, W, P, 1 if neg_abs else 0, _stream(pts))),
                "wm2f_point_sample_levels_fwd")
        ctx.save_for_backward(pts, index)
        ctx.shape = (NL, N, H, W)
        ctx.unique = bool(unique) and W <= 16384
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        pts, index = ctx.saved_tensors
        NL, N, H, W = ctx.shape
        grad_out = _req(grad_out, "grad_out")
        grads = [torch.zeros(N, H, W, device=pts.device, dtype=torch.float32) for _ in range(NL)]
        fn = load().wm2f_point_sample_levels_bwd_unique if ctx.unique else load().wm2f_point_sample_levels_bwd
        with torch.cuda.device(pts.device):
            check(_timed("point_sample_levels_bwd", pts, lambda: fn(
                _p(grad_out), _p(pts), _p(index), _ptr_table(grads), NL, pts.shape[1], H, W, pts.shape[2], _stream(pts))),
                "wm2f_point_sample_levels_bwd")
        return (None, None, None, None, *grads)


def select_top_points(score: torch.Tensor, pts: torch.Tensor, k: int, out_points: int | None = None) -> torch.Tensor:
    """The points of the k largest scores of each row (HF:688-704: `gather(coords, topk(uncertainty, k)[1])`), without the sort
    a stock top-k of thousands is: score (R, n) fp32, pts (R, n, 2) -> (R, out_points or k, 2) whose first k entries are the
    selected points in INDEX order (the losses sum over points: only the set matters; equal scores at the threshold: lowest
    indices first; NaN ranks highest).  Entries k.. are left for the caller (the random points of HF:700-703).  No autograd."""
    score, pts = _req(score, "score"), _req(pts, "pts")
    R, n = score.shape
    if pts.shape != (R, n, 2) or not 0 < k <= n:
        raise ValueError(f"select_top_points: score {tuple(score.shape)} pts {tuple(pts.shape)} k {k}")
    P = int(out_points) if out_points is not None else int(k)
    out = torch.empty(R, P, 2, device=score.device, dtype=torch.float32)
    with torch.cuda.device(score.device):
        rc = _timed("select_top_points", score, lambda: load().wm2f_select_top_points(_p(score), _p(pts), _p(out), R, n, int(k), P, _stream(score)))
    if rc == _lib.WM2F_EUNSUPPORTED:  # more candidates per row than LDS holds: the stock sort
        idx = torch.sort(torch.topk(score, k=k, dim=1)[1], dim=1)[0]  # index order, as the kernel writes them
        out[:, :k] = torch.gather(pts, 1, idx[..., None].expand(-1, -1, 2))
        return out
    check(rc, "wm2f_select_top_points")
    return out


def point_sample_levels(maps, pts: torch.Tensor, index: torch.Tensor, neg_abs: bool = False, unique_index: bool = False) -> torch.Tensor:
    """sample_point (HF:245-274) on one (N,H,W) map tensor PER LEVEL without stacking them: pts (NL,M,P,2) in [0,1]
    (x,y), index (NL,M) int32 = which map of its level row m samples -> (NL,M,P).  neg_abs: -|value| (HF:688-690).
    unique_index: the caller guarantees that no map is indexed twice within a level (the matched rows of a one-to-one
    assignment) -- the backward then accumulates each map in LDS bands and stores it, without global atomics."""
    return _PointSampleLevels.apply(pts, index, bool(neg_abs), bool(unique_index), *maps)


class _MaskLossRows(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, logits, labels):
        logits, labels = _req(logits, "logits"), _req(labels, "labels")
        R, P = logits.shape
        sums = torch.empty(R, 4, device=logits.device, dtype=torch.float32)
        bce, dice = torch.empty(R, device=logits.device), torch.empty(R, device=logits.device)
        with torch.cuda.device(logits.device):
            check(_timed("mask_loss_rows_fwd", logits, lambda: load().wm2f_mask_loss_rows_fwd(
                _p(logits), _p(labels), _p(sums), _p(bce), _p(dice), R, P, _stream(logits))), "wm2f_mask_loss_rows_fwd")
        ctx.save_for_backward(logits, labels, sums)
        return bce, dice

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_bce, g_dice):
        logits, labels, sums = ctx.saved_tensors
        R, P = logits.shape
        g_bce, g_dice = _req(g_bce, "g_bce"), _req(g_dice, "g_dice")
        grad = torch.empty_like(logits)
        with torch.cuda.device(logits.device):
            check(_timed("mask_loss_rows_bwd", logits, lambda: load().wm2f_mask_loss_rows_bwd(
                _p(logits), _p(labels), _p(sums), _p(g_bce), _p(g_dice), _p(grad), R, P, _stream(logits))),
                "wm2f_mask_loss_rows_bwd")
        return grad, None


def mask_loss_rows(logits: torch.Tensor, labels: torch.Tensor):
    """Per matched mask (row): (mean BCE-with-logits over its points HF:308-324, dice HF:278-305), differentiable in logits."""
    return _MaskLossRows.apply(logits, labels)


# ------------------------------------------------------------------------- label expansion
def labelmap_to_masks(label_map: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """(H, W) int32 id map, (T,) int32 ids -> (T, H, W) uint8 masks `label_map == ids[t]` on the device."""
    label_map, ids = _req(label_map, "label_map", torch.int32), _req(ids, "ids", torch.int32)
    H, W = label_map.shape
    T = int(ids.shape[0])
    out = torch.empty(T, H, W, device=label_map.device, dtype=torch.uint8)
    if T == 0:
        return out
    if (H * W) % 4:
        raise ValueError("labelmap_to_masks: H * W must be divisible by 4")
    with torch.cuda.device(label_map.device):
        check(load().wm2f_labelmap_to_masks(_p(label_map), _p(ids), _p(out), H * W, T, _stream(label_map)),
              "wm2f_labelmap_to_masks")
    return out
