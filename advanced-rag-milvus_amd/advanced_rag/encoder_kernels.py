"""Weight packing and launch helpers for the hand-written encoder layer kernels (csrc/encoder_layer.h).

The forward passes behind the reference's plugin hooks — `CrossEncoderReranker.model.predict` (reference
retrieval.py:651-685) and `embedding_generator.encode_semantic` (indexing.py:610-620) — are post-LN BERT layers.  On the
GPU a layer runs as three launches: hr_linear_rows_f16_dev (QKV projection) -> hr_attention_f16_dev ->
hr_encoder_tail_f16_dev (output projection + residual + LayerNorm + FFN + residual + LayerNorm).  The kernels take their
weights as streams of 1 KiB pieces, one MFMA A fragment each (16 output features x 32 inputs, register-image order: lane
16 g + r holds 8 consecutive halves of row r), built here ONCE per model from the nn.Linear weights:

  natural k order      lane (g, r), element j  <-  W[16 t + r][32 s + 8 g + j]
  accumulator k order  lane (g, r), element j  <-  W[16 t + r][32 s + (4 g + j if j < 4 else 16 + 4 g + j - 4)]
                       (the operand of that product is the previous product's accumulator, whose lane holds output rows
                       4 g .. 4 g + 3 of two neighbouring 16-row tiles)

Pure tensor reshapes: they run wherever the weights live (tests check them on the CPU against the index formulas).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native

SUPPORTED_TAIL = {(384, 1536)}     # (hidden, intermediate) the fused tail kernel is instantiated for
SUPPORTED_LINEAR_K = {384}


def pack_natural(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> pieces [N/16][K/32] x (4 g, 16 r, 8 j), natural k order; returns a flat fp16 tensor."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    v = w.to(torch.float16).reshape(N // 16, 16, K // 32, 4, 8)          # [t][r][s][g][j]
    return v.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)              # [t][s][g][r][j]


def pack_accumulator_order(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> pieces [N/16][K/32] x (4 g, 16 r, 8 j) with k = 32 s + 16 (j >> 2) + 4 g + (j & 3)."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    v = w.to(torch.float16).reshape(N // 16, 16, K // 32, 2, 4, 4)        # [t][r][s][hi][g][jj]
    return v.permute(0, 2, 4, 1, 3, 5).contiguous().reshape(-1)           # [t][s][g][r][hi][jj] -> j = 4 hi + jj


def store_order_rows(n: int, device=None) -> torch.Tensor:
    """Row permutation of a weight matrix (and its bias) for hr_linear_rows_f16_dev: packed row 32 so + 16 u + r holds output
    feature 32 so + 8 (r >> 2) + 4 u + (r & 3), so that a lane's eight results of a stage are eight consecutive features."""
    assert n % 32 == 0
    i = torch.arange(n, device=device)
    so, u, r = i // 32, (i // 16) % 2, i % 16
    return 32 * so + 8 * (r // 4) + 4 * u + (r % 4)


def pack_linear(w: torch.Tensor, bias: torch.Tensor, fragment_order_input: bool):
    """-> (packed weights, fp32 bias) of hr_linear_rows_f16_dev for row-major (natural k order) or fragment-order input."""
    perm = store_order_rows(w.shape[0], w.device)
    wp = w[perm]
    return (pack_accumulator_order(wp) if fragment_order_input else pack_natural(wp)), bias[perm].to(torch.float32).contiguous()


def pack_tail_stream(w_out: torch.Tensor, w_up: torch.Tensor, w_down: torch.Tensor) -> torch.Tensor:
    """The weight stream of hr_encoder_tail_f16_dev (include/hbmrag.h): W_out | up(0) | up(1) down(0) | ... | down(last)."""
    H, I = w_out.shape[0], w_up.shape[0]
    assert w_out.shape == (H, H) and w_up.shape == (I, H) and w_down.shape == (H, I) and H % 32 == 0 and I % 64 == 0
    piece = 512                                                            # halves per piece
    sp = H // 16                                                           # pieces per stage
    out_p = pack_accumulator_order(w_out).reshape(-1, sp * piece)          # H/32 stages of two output tiles (the attention
                                                                           # output arrives in fragment order = accumulator k order)
    up_p = pack_accumulator_order(w_up).reshape(I // 32, sp * piece)      # stage c = tiles 2c, 2c+1, all k-steps
    down_p = pack_accumulator_order(w_down).reshape(H // 16, I // 32, piece).permute(1, 0, 2).reshape(I // 32, sp * piece)
    n = I // 32
    stages = [out_p, up_p[0:1]]
    inter = torch.stack([up_p[1:], down_p[:-1]], dim=1).reshape(2 * (n - 1), sp * piece)   # up(c+1), down(c)
    stages += [inter, down_p[-1:]]
    return torch.cat(stages).reshape(-1).contiguous()


def tail_tables(b_out, g1, be1, b_down, g2, be2, b_up) -> torch.Tensor:
    return torch.cat([t.reshape(-1).to(torch.float32) for t in (b_out, g1, be1, b_down, g2, be2, b_up)]).contiguous()


class LayerKernels:
    """The packed operands of one encoder layer on the GPU, rebuilt when a parameter's version counter moves."""

    def __init__(self):
        self._key = None
        self.qkv_w = self.qkv_w_fr = self.qkv_b = self.kv_w = self.kv_w_fr = self.kv_b = self.stream = self.tables = None

    @staticmethod
    def supports(layer) -> bool:
        H, I = layer.out.weight.shape[0], layer.up.weight.shape[0]
        return (H, I) in SUPPORTED_TAIL and H in SUPPORTED_LINEAR_K

    def ensure(self, layer) -> "LayerKernels":
        params = (layer.qkv.weight, layer.qkv.bias, layer.out.weight, layer.out.bias, layer.ln1.weight, layer.ln1.bias,
                  layer.up.weight, layer.up.bias, layer.down.weight, layer.down.bias, layer.ln2.weight, layer.ln2.bias)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key:
            H = layer.out.weight.shape[0]
            with torch.no_grad():
                self.qkv_w, self.qkv_b = pack_linear(layer.qkv.weight, layer.qkv.bias, False)      # row-major input
                self.qkv_w_fr, _ = pack_linear(layer.qkv.weight, layer.qkv.bias, True)             # fragment-order input
                # keys and values only (the last layer of a cross-encoder)
                self.kv_w, self.kv_b = pack_linear(layer.qkv.weight[H:], layer.qkv.bias[H:], False)
                self.kv_w_fr, _ = pack_linear(layer.qkv.weight[H:], layer.qkv.bias[H:], True)
                self.stream = pack_tail_stream(layer.out.weight, layer.up.weight, layer.down.weight)
                self.tables = tail_tables(layer.out.bias, layer.ln1.weight, layer.ln1.bias, layer.down.bias,
                                          layer.ln2.weight, layer.ln2.bias, layer.up.bias)
            self._key = key
        return self


def fr_rows(rows: int) -> int:
    """Rows a fragment-order buffer holds: whole 16-row tiles."""
    return -(-rows // 16) * 16


def to_fragment_order(x2d: torch.Tensor) -> torch.Tensor:
    """[M, H] row-major -> the fragment-order buffer the kernels exchange (include/hbmrag.h); pure reshapes (tests, tools)."""
    M, H = x2d.shape
    pad = torch.zeros((fr_rows(M), H), dtype=x2d.dtype, device=x2d.device)
    pad[:M] = x2d
    v = pad.reshape(-1, 16, H // 32, 2, 4, 4)                     # [tile][col][s][hi][g][jj]
    return v.permute(0, 2, 4, 1, 3, 5).contiguous().reshape(fr_rows(M), H)   # [tile][s][g][col][hi][jj]


def from_fragment_order(x_fr: torch.Tensor, rows: int) -> torch.Tensor:
    H = x_fr.shape[1]
    v = x_fr.reshape(-1, H // 32, 4, 16, 2, 4)                    # [tile][s][g][col][hi][jj]
    return v.permute(0, 3, 1, 4, 2, 5).contiguous().reshape(-1, H)[:rows]


def linear_rows(x2d: torch.Tensor, w_packed: torch.Tensor, bias_f32: torch.Tensor, n_out: int, rows: Optional[int] = None,
                x_fr: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x2d [M, K] fp16 (row-major; or fragment order with `rows` valid rows) -> [rows, n_out] fp16 row-major on the current stream."""
    M = x2d.shape[0] if rows is None else rows
    K = x2d.shape[1]
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=x2d.device)
    _native.linear_rows_f16_dev(x2d.data_ptr(), x_fr, w_packed.data_ptr(), bias_f32.data_ptr(), out.data_ptr(), M, K, n_out,
                                out.stride(0), torch.cuda.current_stream(x2d.device).cuda_stream)
    return out


def encoder_tail(attn_fr: torch.Tensor, x2d: torch.Tensor, k: LayerKernels, intermediate: int, eps: float, gelu_erf: bool,
                 rows: Optional[int] = None, x_fr: bool = False, out_fr: bool = False) -> torch.Tensor:
    """attn_fr: the attention output in fragment order; x2d: the layer input, row-major [rows, H] or fragment order;
    -> the layer output, row-major [rows, H] or fragment order [fr_rows(rows), H]."""
    M = x2d.shape[0] if rows is None else rows
    H = x2d.shape[1]
    out = torch.empty((fr_rows(M) if out_fr else M, H), dtype=torch.float16, device=x2d.device)
    _native.encoder_tail_f16_dev(attn_fr.data_ptr(), x2d.data_ptr(), x_fr, out.data_ptr(), out_fr, k.stream.data_ptr(),
                                 k.tables.data_ptr(), M, H, intermediate, eps, gelu_erf,
                                 torch.cuda.current_stream(x2d.device).cuda_stream)
    return out
